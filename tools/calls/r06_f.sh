# diagnosis: the GoPro certified pass is right on the interpreter (0 wrong certificates, bit-exact) and wrong on the GPU for bilinear builds
timeout 300 python3 tools/diag_gopro.py 2>&1 | grep -v amdgpu.ids | tee -a $O/summary.txt
GFW_JIT_WAVES=6 timeout 300 python3 tools/diag_gopro.py 2>&1 | grep -v amdgpu.ids | head -5 | tee -a $O/summary.txt
GFW_JIT_DEFS="-O1" timeout 300 python3 tools/diag_gopro.py 2>&1 | grep -v amdgpu.ids | head -5 | tee -a $O/summary.txt
