for s in 4 3 2 1; do echo "SHK_ASM_SLICES=$s"; SHK_ASM_SLICES=$s python tools/probe_kernels.py c4_10m 2>&1 | grep -E "setup|assemble" | cut -c1-400; done
