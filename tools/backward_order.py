"""Prints the order in which one eager backward pass issues its contractions and stream events (which stream, which
event), to read a step timeline against.  Measurement aid."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench import u20k_cloud, synthetic_targets
from lisec_amd import Constants, _lib, ops
from lisec_amd.network import LisecNet
from lisec_amd.voxelizer import Voxelizer

dev = torch.device("cuda")
net = LisecNet(Constants.nx, Constants.ny, Constants.nz, Constants.maxPoints, device=dev)
vox = Voxelizer(Constants.voxelx, Constants.voxely, Constants.voxelz, Constants.maxPoints, Constants.nx // 2, Constants.ny // 2,
                Constants.nz, device=dev)
pts = torch.from_numpy(u20k_cloud(0)).to(dev)
yc, yr = synthetic_targets(0, net.Ho, net.Wo)
yc, yr = torch.from_numpy(yc).to(dev), torch.from_numpy(yr).to(dev)
for _ in range(2):
    net.train_step(vox(pts), yc, yr)
torch.cuda.synchronize()

main = torch.cuda.current_stream().cuda_stream
pinned = [main]
names = {}
real_pin = _lib.pin_stream


def pin(h):
    prev = real_pin(h)
    pinned[0] = h
    return prev


_lib.pin_stream = pin
import lisec_amd.network as nw


def sname(h):
    return "main" if h == main else ("side" if h == net.side.cuda_stream else hex(h or 0))


def ename(ev):
    if id(ev) not in names:
        for k, v in net._fwd_events.items():
            if v is ev:
                names[id(ev)] = k
        names.setdefault(id(ev), "ev%d" % len(names))
    return names[id(ev)]


LisecNet._record = staticmethod(lambda ev, s: (print(f"   record {ename(ev)} on {sname(s.cuda_stream)}"), ev.record(s.cuda_stream))[1])
LisecNet._wait = staticmethod(lambda ev, s: (print(f"   {sname(s.cuda_stream)} waits {ename(ev)}"), ev.wait(s.cuda_stream))[1])
for fn in ("conv_forward", "conv_wgrad", "head_compose_backward", "colsum", "bn_backward_apply_coef", "bn_backward"):
    real = getattr(ops, fn)

    def wrapped(*a, _real=real, _fn=fn, **k):
        g = a[0]
        desc = f"M={g.M} Cin={g.Cin} Cout={g.Cout}" if hasattr(g, "M") else ""
        print(f"{sname(pinned[0]):5s} {_fn} {desc}")
        return _real(*a, **k)

    setattr(ops, fn, wrapped)

net.forward(vox(pts), training=True)
print("---- backward ----")
net.backward(yc, yr)
torch.cuda.synchronize()
