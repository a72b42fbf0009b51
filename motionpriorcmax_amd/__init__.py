"""motionpriorcmax_amd -- MI355X-native contrast-maximisation (CMax) loss hot path.

Drop-in for the loss plugin API of tub-rip/MotionPriorCMax (`src/losses`): the numerical path
runs in hand-written HIP kernels for gfx950 (libmpcmax.so, C ABI in include/mpcmax.h)."""
from .losses import FocusLoss, LossFactory, TrajectoryLossBase  # noqa: F401

__all__ = ['LossFactory', 'TrajectoryLossBase', 'FocusLoss']
