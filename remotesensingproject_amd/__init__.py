"""MI355X-native EPI depth scan (drop-in for RSLightFields' Depth1DComputer_pile path).

The compute path is the HIP library ``librslf_hip.so`` (csrc/), reached through
the C-ABI declared in ``include/rslf_hip.h``.  There is no CPU fallback: every
entry point raises if the library is missing.
"""
__version__ = "0.1.0"
