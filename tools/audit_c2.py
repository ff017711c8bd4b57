import sys; sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
from gyroflow_amd import synthetic as S
import test_gpu_pass1 as T
fr = S.SyntheticFrame("YUV422P16LE", 3840, 2160, seed=0x9F10)
(c,w,q,o,g),_ = T.audit(fr)
print("certified",c,"wrong",w,"queued",q,"overflow",o,"queued frac",(q+o)/(3840*2160),"max gap px",g)
