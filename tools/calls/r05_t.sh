# RGBAf16 on the fused kernel; the whole GPU suite
bench A=1 --fmt RGBAF16 --steps 100
bench A=1 --fmt RGBAF16 --steps 100 --jit 0 --clip 1
bench GFW_UNUSED=1 --fmt RGBAF16 --steps 50 --variant 1 --clip 1
timeout 1500 python3 -m pytest tests -x -q -m gpu 2>&1 | tail -5
