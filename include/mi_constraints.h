/*
 * mi_constraints.h — joint descriptors for mi_constraint_create / mi_constraint_update.
 *
 * Field-for-field the reference's constraint structs (src/physics/constraints.h), as plain
 * 4-byte-packed floats: quaternions are x,y,z,w; the anonymous motor unions are single floats;
 * constraint_motor_type is a uint32 (0 = velocity motor, 1 = position motor,
 * src/physics/constraints.h:41-45).
 *
 * Reference layout.  The structs below ARE the reference's structs as they lie in memory, except for the 8 bytes
 * of tail padding that a leading quat (16-byte aligned) gives fixed_constraint (40 -> 48 B) and slider_constraint
 * (72 -> 80 B).  mi_constraint_create / _update / _get / mi_constraints_update therefore accept `bytes` = the packed size
 * OR the reference's sizeof (MI_REF_SIZEOF_*): a reference-side caller passes `&constraint, sizeof(constraint)` of its own
 * struct, no member-wise conversion (the padding is ignored on the way in and left untouched on the way out).
 */
#ifndef MI_CONSTRAINTS_H
#define MI_CONSTRAINTS_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

enum { MI_MOTOR_VELOCITY = 0, MI_MOTOR_POSITION = 1 };
/* sizeof of the reference's structs (src/physics/constraints.h:73-80, 129-135, 175-183, 229-257, 346-380, 497-520) */
enum { MI_REF_SIZEOF_DISTANCE_CONSTRAINT = 28, MI_REF_SIZEOF_BALL_CONSTRAINT = 24, MI_REF_SIZEOF_FIXED_CONSTRAINT = 48, MI_REF_SIZEOF_HINGE_CONSTRAINT = 104,
       MI_REF_SIZEOF_CONE_TWIST_CONSTRAINT = 120, MI_REF_SIZEOF_SLIDER_CONSTRAINT = 80 };

/* distance_constraint — src/physics/constraints.h:73-80 */
typedef struct mi_distance_constraint {
    float local_anchor_a[3];
    float local_anchor_b[3];
    float global_length;
} mi_distance_constraint;

/* ball_constraint — src/physics/constraints.h:129-135 */
typedef struct mi_ball_constraint {
    float local_anchor_a[3];
    float local_anchor_b[3];
} mi_ball_constraint;

/* fixed_constraint — src/physics/constraints.h:175-183 */
typedef struct mi_fixed_constraint {
    float initial_inv_rotation_difference[4];
    float local_anchor_a[3];
    float local_anchor_b[3];
} mi_fixed_constraint;

/* hinge_constraint — src/physics/constraints.h:229-257 */
typedef struct mi_hinge_constraint {
    float local_anchor_a[3];
    float local_anchor_b[3];
    float local_hinge_axis_a[3];
    float local_hinge_axis_b[3];
    float min_rotation_limit;   /* [-pi, 0], else disabled */
    float max_rotation_limit;   /* [0, pi], else disabled */
    float max_motor_torque;
    uint32_t motor_type;
    float motor_velocity_or_target_angle;
    float local_hinge_tangent_a[3];
    float local_hinge_bitangent_a[3];
    float local_hinge_tangent_b[3];
} mi_hinge_constraint;

/* cone_twist_constraint — src/physics/constraints.h:346-380 */
typedef struct mi_cone_twist_constraint {
    float local_anchor_a[3];
    float local_anchor_b[3];
    float local_limit_axis_a[3];
    float local_limit_axis_b[3];
    float local_limit_tangent_a[3];
    float local_limit_bitangent_a[3];
    float local_limit_tangent_b[3];
    float swing_limit;
    float twist_limit;
    uint32_t swing_motor_type;
    float swing_motor_velocity_or_target_angle;
    float max_swing_motor_torque;
    float swing_motor_axis;
    uint32_t twist_motor_type;
    float twist_motor_velocity_or_target_angle;
    float max_twist_motor_torque;
} mi_cone_twist_constraint;

/* slider_constraint — src/physics/constraints.h:497-520 */
typedef struct mi_slider_constraint {
    float initial_inv_rotation_difference[4];
    float local_anchor_a[3];
    float local_anchor_b[3];
    float local_axis_a[3];
    float neg_distance_limit;
    float pos_distance_limit;
    float max_motor_force;
    uint32_t motor_type;
    float motor_velocity_or_target_distance;
} mi_slider_constraint;

#ifdef __cplusplus
}
#endif
#endif
