"""Probe for the planned peer-copy gather: does a device-to-device copy issued on another stream run
beside a k_pass launch that holds every wave slot, or does it wait for the launch to drain?"""
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import rustracer_amd as ra
from rustracer_amd import _abi
from rustracer_amd.scenes import demo_camera, demo_world
from rustracer_amd.world import flatten_camera

W, H, NF = 1920, 1080, 60
world = demo_world()
cam = flatten_camera(demo_camera(W, H))
rr = ra.PathTracer(world, ra.BLACK, 10, 3, 3).flatten()
dev = world.device()
dev.set_schedule((0,))
lib = _abi.lib()
img = torch.zeros((NF, H, W, 3), device="cuda")
cams = (_abi.rt_camera * NF)(*([cam] * NF))
outs = (C.c_void_p * NF)(*[img[b].data_ptr() for b in range(NF)])
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
src = torch.ones((64 << 20) // 4, device="cuda")
dst = torch.zeros_like(src)


def render():
    _abi.check(lib.rt_render_frames(dev.handle, cams, NF, C.byref(rr), W, H, 42, 54, 1, None, outs, sa.cuda_stream, None))


render()
torch.cuda.synchronize()
for what in ("copy_ (blit / SDMA)", "elementwise kernel (mul)"):
    r0, r1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    c0, c1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    r0.record(sa)
    render()
    r1.record(sa)
    # give the launch a moment to take the whole GPU, then issue the copy on the other stream
    time.sleep(0.01)
    with torch.cuda.stream(sb):
        c0.record(sb)
        if what.startswith("copy_"):
            dst.copy_(src, non_blocking=True)
        else:
            torch.mul(src, 2.0, out=dst)
        c1.record(sb)
    torch.cuda.synchronize()
    print(f"{what:28s}: render {r0.elapsed_time(r1):7.2f} ms; 64 MiB copy issued during it took {c0.elapsed_time(c1):7.3f} ms "
          f"(render start -> copy end {r0.elapsed_time(c1):7.2f} ms)", flush=True)
