import sys; sys.path.insert(0,'.')
import numpy as np
from raytracing_amd import rt_bench as rb
from oracle import rt_oracle as O
for sc,lim in (("fisheye",(-1.5,1.5,-1.5,1.5)),("interface",(-2,20,-2,4))):
    F=rb.Field.build(sc); OF=O.Field(sc,lim,rb.DELTA)
    x,y,Z,cdy,cdx=F.arrays(); ox,oy,oZ,ocdy,ocdx=OF.arrays()
    d=np.abs(Z-oZ); i,j=np.unravel_index(np.argmax(d),d.shape)
    print(sc,"max diff",d.max(),"at",i,j,"x,y=",x[j],y[i],"Z",repr(Z[i,j]),repr(oZ[i,j]), "frac differing", np.mean(d>0))
    u=np.abs(Z-oZ)/np.spacing(np.abs(oZ)); print(" ulp hist", np.bincount(np.minimum(u.astype(int).ravel(),10)))
