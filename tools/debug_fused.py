import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import load_package
from oracle.oracle import OracleStore
pkg = load_package()
old_len=[5,3,4]
n=60
dense=np.arange(1,61).astype(np.float64)
sel=[np.array([1,4,3,1],np.int32), np.array([0,1,-1],np.int32), np.array([2,0,3],np.int32)]
mid=[4,3,3]
for axis_map in ([0,0,1,1],[0,1,0,1]):
    maps=[np.array(axis_map,np.uint32), np.arange(3,dtype=np.uint32), np.arange(3,dtype=np.uint32)]
    new=[2,3,3]
    for method in ("first","sum"):
        o=OracleStore(n,"float32",0.0); o.set_data(dense)
        ev,_=o.dice(old_len,mid,sel).drill_up(mid,new,maps,method).typed()
        g=pkg.HipStore(n,"float32",0.0); g.set_data_f64(dense)
        f=g.dice_drillup(old_len,mid,new,sel,maps,method).get_data()
        t=g.dice(old_len,mid,sel).drill_up(mid,new,maps,method).get_data()
        print(axis_map, method); print(' oracle',ev); print(' fused ',f); print(' 2step ',t)
