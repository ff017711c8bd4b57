import sys,os,ctypes as C,subprocess
sys.path.insert(0,'tools'); sys.path.insert(0,'.')
import build_jit_cache as B
from gyroflow_amd import abi
lib=abi.load_library()
defs,header,name=B.key_of(lib,B.bench_frame(interp=4))
lib.gfw_debug_jit_compile.argtypes=[C.c_char_p]*4+[C.c_char_p,C.c_size_t]; lib.gfw_debug_jit_compile.restype=C.c_long
log=C.create_string_buffer(1<<16)
out=sys.argv[1]
n=lib.gfw_debug_jit_compile(B.ARCH, defs, header, out.encode(), log, len(log))
print("compiled", n, "bytes; identical to the shipped cache entry:", open(out,'rb').read()==open(os.path.join("gyroflow_amd/jit_cache",name),'rb').read())
