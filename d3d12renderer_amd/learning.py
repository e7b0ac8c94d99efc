"""ctypes binding of libPhysics-Lib.so — the reference's learning DLL ABI (src/learning/learned_locomotion.cpp:395-489,
consumer learning/loco_env.py) plus the batched entry points.  `PhysicsDLL` mirrors the class of the same name in the
reference's loco_env.py (same methods, same return shapes) so that file's LocoEnv works by pointing it at this library;
`BatchedLocoEnv` steps many ragdolls per call."""
import ctypes as C
from pathlib import Path

import numpy as np

LIB_PATH = Path(__file__).resolve().parent / "libPhysics-Lib.so"
_F = C.POINTER(C.c_float)


def _fp(a):
    return a.ctypes.data_as(_F)


class PhysicsDLL:
    def __init__(self, path=None):
        path = Path(path) if path else LIB_PATH
        if not path.exists():
            raise RuntimeError(f"{path} is missing: run `python -m d3d12renderer_amd.build` (there is no CPU fallback)")
        self._physics = C.CDLL(str(path))
        self._physics.updatePhysics.argtypes = (_F, _F, _F)
        self._physics.resetPhysics.argtypes = (_F,)
        self._physics.getPhysicsRanges.argtypes = (_F, _F, _F, _F)
        self._physics.resetPhysicsBatch.argtypes = (C.c_int, _F)
        self._physics.updatePhysicsBatch.argtypes = (_F, _F, _F, C.POINTER(C.c_int))
        self._physics.setPhysicsSeed.argtypes = (C.c_ulonglong,)
        self._physics.setPhysicsDevice.argtypes = (C.c_int,)
        self._physics.getPhysicsError.restype = C.c_char_p
        self.state_size = self._physics.getPhysicsStateSize()
        self.action_size = self._physics.getPhysicsActionSize()

    def ranges(self):
        smin = np.zeros(self.state_size, np.float32); smax = np.zeros(self.state_size, np.float32)
        amin = np.zeros(self.action_size, np.float32); amax = np.zeros(self.action_size, np.float32)
        self._physics.getPhysicsRanges(_fp(smin), _fp(smax), _fp(amin), _fp(amax))
        return smin, smax, amin, amax

    def seed(self, seed):
        self._physics.setPhysicsSeed(C.c_ulonglong(seed))

    def set_device(self, device):
        self._physics.setPhysicsDevice(device)

    def error(self):
        return self._physics.getPhysicsError().decode()

    # --- the reference's single-environment calls
    def reset(self):
        state = np.zeros(self.state_size, np.float32)
        self._physics.resetPhysics(_fp(state))
        return state

    def step(self, action):
        a = np.ascontiguousarray(action, np.float32)
        assert a.shape == (self.action_size,)
        state = np.zeros(self.state_size, np.float32); reward = np.zeros(1, np.float32)
        done = self._physics.updatePhysics(_fp(a), _fp(state), _fp(reward))
        return state, float(reward[0]), done != 0

    # --- batched
    def reset_batch(self, num_envs):
        states = np.zeros((num_envs, self.state_size), np.float32)
        rc = self._physics.resetPhysicsBatch(num_envs, _fp(states))
        if rc != 0:
            raise RuntimeError(f"resetPhysicsBatch failed ({rc}): {self.error()}")
        return states

    def step_batch(self, actions):
        a = np.ascontiguousarray(actions, np.float32)
        n = a.shape[0]
        assert a.shape == (n, self.action_size)
        states = np.zeros((n, self.state_size), np.float32); rewards = np.zeros(n, np.float32); done = np.zeros(n, np.int32)
        rc = self._physics.updatePhysicsBatch(_fp(a), _fp(states), _fp(rewards), done.ctypes.data_as(C.POINTER(C.c_int)))
        if rc != 0:
            raise RuntimeError(f"updatePhysicsBatch failed ({rc}): {self.error()}")
        return states, rewards, done != 0

    def shutdown(self):
        self._physics.shutdownPhysics()


class BatchedLocoEnv:
    """Vectorised counterpart of the reference's LocoEnv (learning/loco_env.py:55-82): `num_envs` ragdolls in one world."""

    def __init__(self, num_envs, path=None, seed=1):
        self.dll = PhysicsDLL(path)
        self.dll.seed(seed)
        self.num_envs = num_envs
        _, _, self.action_min, self.action_max = self.dll.ranges()
        self.states = self.dll.reset_batch(num_envs)

    def reset(self):
        self.states = self.dll.reset_batch(self.num_envs)
        return self.states

    def step(self, actions):
        self.states, rewards, done = self.dll.step_batch(actions)
        return self.states, rewards, done, {}
