from ddrl4nav_amd.data.experience import Experience
from ddrl4nav_amd.data.ring import PinnedRing

__all__ = ["Experience", "PinnedRing"]
