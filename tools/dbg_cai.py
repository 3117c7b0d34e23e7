import sys; sys.path.insert(0,'.')
import numpy as np, torch
from raytracing_amd import rt_bench as rb
fld=rb.Field.build("vert_heterogeneous")
th=np.linspace(0,np.pi/2,1000)
b=rb.Batch(fld,6,rb.DELTA_S,30228,(-2,5,-2.5,1),1,th,-2.0,-2.0,record_stride=64)
b.run()
v=b.view()
class Cai:
    def __init__(self, ptr, shape, typestr):
        self.__cuda_array_interface__={"shape":shape,"typestr":typestr,"data":(ptr,False),"version":2,"strides":None}
t=torch.as_tensor(Cai(v.x,(b.R,),"<f8"),device="cuda")
print(t.device,t.dtype,t.shape, float(t[15]), b.final()[0,15])
s=torch.as_tensor(Cai(v.s_ray,(b.rec_rows,6,b.R),"<f8"),device="cuda")
print(torch.equal(s.cpu(), torch.from_numpy(b.rows())))
