"""MI355X-native actor-learner hot path for DDRL4NAV.

Sub-packages mirror the reference's plug-in surface for this path (USTC_lab.nn / .agent / .data /
.config / .runner); the arithmetic runs in hand-written HIP kernels behind the C ABI declared in
include/ddrl.h (ddrl4nav_amd/csrc -> libddrl_hip.so).  Importing the package is cheap and does not
need a GPU; constructing a net or a HotPath does, and fails loudly without one.
"""
__version__ = "0.1.0"
