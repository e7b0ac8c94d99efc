// ora_cloth.cpp — TEST INFRASTRUCTURE (CPU oracle), see ora_math.h header note.
//
// cloth_component (src/physics/cloth.h:5-60, cloth.cpp:7-352): a particle grid with distance constraints (stretch, shear,
// bend), wind from the global force field, semi-implicit integration and Gauss-Seidel velocity / position / drift passes.
// Stepped after the rigid bodies (physics.cpp:1352-1358); it does not interact with them.
// ORDER_REFERENCE walks the constraints in creation order like the reference.  ORDER_CANONICAL walks them in the device's
// order: 12 colours (family x parity) — constraints of one colour share no particle, so the device solves a colour in
// parallel and the order inside it does not matter.  Both orders are Gauss-Seidel sweeps over the same constraints.
#include "ora_world.h"
#include <algorithm>

namespace ora {

// cloth_component::getParticlePosition — cloth.cpp:126-132
static vec3 particlePosition(const Cloth& c, float relX, float relY) {
    vec3 p(relX * c.width, -relY * c.height, 0.f);
    p.x -= c.width * 0.5f;
    std::swap(p.y, p.z);
    return p;
}
// colour of a constraint from its family (creation slot 0..5: right, down, diagonal, anti-diagonal, right + 2, down + 2) and
// the grid coordinate of the vertex that created it
static uint32_t colourOf(uint32_t family, uint32_t x, uint32_t y) {
    switch (family) {
        case 0: return 0 + (x & 1u);
        case 1: return 2 + (y & 1u);
        case 2: return 4 + (x & 1u);
        case 3: return 6 + (x & 1u);
        case 4: return 8 + ((x >> 1) & 1u);
        default: return 10 + ((y >> 1) & 1u);
    }
}

// cloth_component ctor — cloth.cpp:7-85
Cloth::Cloth(const mi_cloth_desc& d) : width(d.width), height(d.height), gridSizeX(d.grid_size_x), gridSizeY(d.grid_size_y),
                                       totalMass(d.total_mass), gravityFactor(d.gravity_factor), damping(d.damping), stiffness(d.stiffness) {
    uint32_t n = gridSizeX * gridSizeY;
    float invMassPerParticle = (float)n / totalMass;
    velocities.assign(n, vec3(0.f)); forces.assign(n, vec3(0.f));
    for (uint32_t y = 0; y < gridSizeY; ++y) {
        float invMass = (y == 0) ? 0.f : invMassPerParticle;   // upper row locked
        for (uint32_t x = 0; x < gridSizeX; ++x) {
            vec3 p = particlePosition(*this, (float)x / (float)(gridSizeX - 1), (float)y / (float)(gridSizeY - 1));
            positions.push_back(p); prevPositions.push_back(p); invMasses.push_back(invMass);
        }
    }
    auto add = [&](uint32_t a, uint32_t b, uint32_t family, uint32_t x, uint32_t y) {
        constraints.push_back(Constraint{a, b, length(positions[a] - positions[b]), (invMasses[a] + invMasses[b]) / stiffness});
        colours.push_back(colourOf(family, x, y));
    };
    for (uint32_t y = 0; y < gridSizeY; ++y)
        for (uint32_t x = 0; x < gridSizeX; ++x) {
            uint32_t i = y * gridSizeX + x;
            if (x < gridSizeX - 1) add(i, i + 1, 0, x, y);
            if (y < gridSizeY - 1) add(i, i + gridSizeX, 1, x, y);
            if (x < gridSizeX - 1 && y < gridSizeY - 1) { add(i, i + gridSizeX + 1, 2, x, y); add(i + gridSizeX, i + 1, 3, x, y); }
            if (x < gridSizeX - 2) add(i, i + 2, 4, x, y);
            if (y < gridSizeY - 2) add(i, i + gridSizeX * 2, 5, x, y);
        }
    canonicalOrder.resize(constraints.size());
    for (uint32_t i = 0; i < canonicalOrder.size(); ++i) canonicalOrder[i] = i;
    std::stable_sort(canonicalOrder.begin(), canonicalOrder.end(), [&](uint32_t a, uint32_t b) { return colours[a] < colours[b]; });
    oldTotalMass = totalMass; oldStiffness = stiffness;
}

// cloth_component::setWorldPositionOfFixedVertices — cloth.cpp:87-124
void Cloth::setFixedVertices(vec3 tp, quat tr, bool moveRigid) {
    auto xf = [&](vec3 p) { return tr * p + tp; };   // transformPosition, scale 1
    if (moveRigid) {
        vec3 pivot = (gridSizeX % 2 == 1) ? positions[gridSizeX / 2] : (positions[gridSizeX / 2] + positions[gridSizeX / 2 - 1]) * 0.5f;
        vec3 currentAxis = normalize(positions[gridSizeX - 1] - positions[0]);
        vec3 newAxis = normalize(xf(particlePosition(*this, 1.f, 0.f)) - xf(particlePosition(*this, 0.f, 0.f)));
        vec3 newPivot = xf(particlePosition(*this, 0.5f, 0.f));
        quat deltaRotation = rotateFromTo(currentAxis, newAxis);
        for (uint32_t y = 1; y < gridSizeY; ++y)
            for (uint32_t x = 0; x < gridSizeX; ++x) { vec3& p = positions[y * gridSizeX + x]; p = deltaRotation * (p - pivot) + newPivot; }
    }
    for (uint32_t x = 0; x < gridSizeX; ++x) positions[x] = xf(particlePosition(*this, (float)x / (float)(gridSizeX - 1), 0.f));
}

// cloth_component::applyWindForce — cloth.cpp:139-174
void Cloth::applyWindForce(vec3 force) {
    auto normalOf = [](vec3 a, vec3 b, vec3 c) { return cross(b - a, c - a); };
    for (uint32_t y = 0; y < gridSizeY - 1; ++y)
        for (uint32_t x = 0; x < gridSizeX - 1; ++x) {
            uint32_t tl = y * gridSizeX + x, tr = tl + 1, bl = tl + gridSizeX, br = bl + 1;
            {
                vec3 normal = normalOf(positions[tl], positions[bl], positions[tr]);
                vec3 f = normal * dot(normalize(normal), force);
                f *= 1.f / 3.f;
                forces[tl] += f; forces[tr] += f; forces[bl] += f;
            }
            {
                vec3 normal = normalOf(positions[br], positions[tr], positions[bl]);
                vec3 f = normal * dot(normalize(normal), force);
                f *= 1.f / 3.f;
                forces[br] += f; forces[tr] += f; forces[bl] += f;
            }
        }
}

// cloth_component::recalculateProperties — cloth.cpp:299-317
void Cloth::recalculateProperties() {
    float invMassPerParticle = (float)(gridSizeX * gridSizeY) / totalMass;
    for (float& im : invMasses) im = (im != 0.f) ? invMassPerParticle : 0.f;
    stiffness = clampf(stiffness, 0.01f, 1.f);
    float invStiffness = 1.f / stiffness;
    for (Constraint& c : constraints) c.inverseMassSum = (invMasses[c.a] + invMasses[c.b]) * invStiffness;
}

// cloth_component::simulate / solveVelocities / solvePositions — cloth.cpp:182-297
void Cloth::simulate(uint32_t velocityIterations, uint32_t positionIterations, uint32_t driftIterations, float dt, bool canonical) {
    if (totalMass != oldTotalMass || stiffness != oldStiffness) { recalculateProperties(); oldTotalMass = totalMass; oldStiffness = stiffness; }
    const uint32_t n = gridSizeX * gridSizeY, nc = (uint32_t)constraints.size();
    auto at = [&](uint32_t k) { return canonical ? canonicalOrder[k] : k; };
    float gravityVelocity = -9.81f * dt * gravityFactor;
    for (uint32_t i = 0; i < n; ++i) {
        if (invMasses[i] > 0.f) velocities[i].y += gravityVelocity;
        velocities[i] += forces[i] * (invMasses[i] * dt);
        prevPositions[i] = positions[i];
        positions[i] += velocities[i] * dt;
        forces[i] = vec3(0.f);
    }
    float invDt = (dt > 1e-5f) ? (1.f / dt) : 1.f;
    auto solvePositions = [&]() {
        for (uint32_t k = 0; k < nc; ++k) {
            const Constraint& c = constraints[at(k)];
            if (c.inverseMassSum > 0.f) {
                vec3 delta = positions[c.b] - positions[c.a];
                float len = squaredLength(delta);
                float sqRest = c.restDistance * c.restDistance;
                if (sqRest + len > 1e-5f) {
                    float kk = ((sqRest - len) / (c.inverseMassSum * (sqRest + len)));
                    positions[c.a] -= delta * (kk * invMasses[c.a]);
                    positions[c.b] += delta * (kk * invMasses[c.b]);
                }
            }
        }
    };
    if (velocityIterations > 0) {
        std::vector<vec3> gradient(nc); std::vector<float> inverseScaledGradientSquared(nc);
        for (uint32_t i = 0; i < nc; ++i) {
            const Constraint& c = constraints[i];
            gradient[i] = prevPositions[c.b] - prevPositions[c.a];
            inverseScaledGradientSquared[i] = (c.inverseMassSum == 0.f) ? 0.f : (1.f / (squaredLength(gradient[i]) * c.inverseMassSum));
        }
        for (uint32_t it = 0; it < velocityIterations; ++it)
            for (uint32_t k = 0; k < nc; ++k) {
                uint32_t i = at(k);
                const Constraint& c = constraints[i];
                float j = -dot(gradient[i], velocities[c.a] - velocities[c.b]) * inverseScaledGradientSquared[i];
                velocities[c.a] += gradient[i] * (j * invMasses[c.a]);
                velocities[c.b] -= gradient[i] * (j * invMasses[c.b]);
            }
        for (uint32_t i = 0; i < n; ++i) positions[i] = prevPositions[i] + velocities[i] * dt;
    }
    if (positionIterations > 0) {
        for (uint32_t it = 0; it < positionIterations; ++it) solvePositions();
        for (uint32_t i = 0; i < n; ++i) velocities[i] = (positions[i] - prevPositions[i]) * invDt;
    }
    if (driftIterations > 0) {
        for (uint32_t i = 0; i < n; ++i) prevPositions[i] = positions[i];
        for (uint32_t it = 0; it < driftIterations; ++it) solvePositions();
        for (uint32_t i = 0; i < n; ++i) velocities[i] += (positions[i] - prevPositions[i]) * invDt;
    }
    float dampingFactor = 1.f / (1.f + dt * damping);
    for (uint32_t i = 0; i < n; ++i) velocities[i] *= dampingFactor;
}

}  // namespace ora
