from ddrl4nav_amd.data.experience import Experience
from ddrl4nav_amd.data.ring import PinnedRing
from ddrl4nav_amd.data.easybytes import EasyBytes

__all__ = ["Experience", "PinnedRing", "EasyBytes"]
