import sys, time, numpy as np
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import torch
from dolfinx_eqlb_amd import cpp, distributed as dd
from synthetic import make_compatible_data
part = dd.StripPartition(500, 0, 1); mesh = part.mesh; ft = part.facet_types()
k = 2
G, f = make_compatible_data(mesh, k, ft, seed=1)
dev = torch.device("cuda:0")
dm = cpp.DeviceMesh(mesh)
dG = torch.from_numpy(G).to(dev); df = torch.from_numpy(f).to(dev)
for out in (0, 1):
    eq = cpp.ConstrainedMinEquilibrator(dm, k, 1)
    eq.set_option("output", out)
    eq.set_boundary(ft)
    n = eq.ndofs if out == 0 else mesh.ncells * 8
    dx = torch.zeros(n, dtype=torch.float64, device=dev)
    s = torch.cuda.current_stream().cuda_stream
    for _ in range(5):
        eq.equilibrate_device(dG.data_ptr(), df.data_ptr(), dx.data_ptr(), s)
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50):
        eq.equilibrate_device(dG.data_ptr(), df.data_ptr(), dx.data_ptr(), s)
    e1.record(); torch.cuda.synchronize()
    print("EV k=2 output", out, "ms/step", e0.elapsed_time(e1) / 50)
