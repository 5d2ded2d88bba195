import sys, time
sys.path.insert(0, ".")
import numpy as np, snn_amd
from snn_amd import synthetic
dn = snn_amd.DeviceNetwork(model=snn_amd.HODGKIN_HUXLEY, nt_kinetics=snn_amd.NT_DESTEXHE, receptor_kinetics=snn_amd.RC_DESTEXHE)
n=128*128
dn.add_lattice(0,128,128); dn.finalize()
dn.set_attr(0,"current_voltage", synthetic.uniform(3,n,-70.0,-60.0))
fl=np.zeros((n,3),np.uint32); fl[:,0]=1
dn.set_attr(0,"neurotransmitters$flags",fl); dn.set_attr(0,"receptors$flags",fl)
dn.fill_graph_synthetic(4,0.5,1.5,with_diagonal=False); dn.set_synapses(True,True)
dn.run(50)
ts=[]
for _ in range(7):
    t0=time.perf_counter(); dn.run(500); ts.append((time.perf_counter()-t0)/500*1e6)
print("c3 us/step", [round(t,1) for t in sorted(ts)])
