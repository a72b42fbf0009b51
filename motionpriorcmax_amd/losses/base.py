from abc import ABC, abstractmethod


class TrajectoryLossBase(ABC):
    """Loss plugin interface; mirrors reference src/losses/base.py:4-14."""

    def __init__(self) -> None:
        self.is_needing_offsets = None

    @abstractmethod
    def get_reconstruction_times(self, device):
        pass

    @abstractmethod
    def calc(self, trajectories, times, **kwargs):
        pass
