"""CPU emulation tier (test infrastructure only): see tests/emu/shim/hip/hip_runtime.h."""
