#!/usr/bin/env python3
"""What the first host -> device transfer of a process pays (GPU box, fresh process): python tools/first_transfer_probe.py"""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from graphpope_amd import _lib, engine, synth
lib = _lib.load()
t = time.perf_counter
ms = lambda a: (t() - a) * 1e3
t0 = t(); dev = engine.require_gpu(); torch.cuda.current_stream().synchronize(); print(f"require_gpu + stream sync            {ms(t0):8.2f} ms")
ei_np, n = synth.flickr_like()
ei = torch.as_tensor(ei_np)
nbytes = ei.numel() * 8
t0 = t(); rc = lib.pope_host_pin(_lib.ptr(ei), nbytes); print(f"hipHostRegister 14 MB (rc {rc})         {ms(t0):8.2f} ms")
t0 = t(); out = torch.empty(ei.shape, dtype=ei.dtype, device=dev); print(f"torch.empty on the device (first)     {ms(t0):8.2f} ms")
t0 = t(); torch.cuda.current_stream().synchronize(); print(f"stream sync                           {ms(t0):8.2f} ms")
import ctypes
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
t0 = t(); _lib.check(lib.pope_copy_to_device(_lib.ptr(ei), _lib.ptr(out), nbytes, st)); print(f"hipMemcpyAsync H2D enqueue (first)    {ms(t0):8.2f} ms")
t0 = t(); torch.cuda.current_stream().synchronize(); print(f"  ... until it has completed          {ms(t0):8.2f} ms")
t0 = t(); ev = torch.cuda.Event(); ev.record(); ev.synchronize(); print(f"first event record + sync             {ms(t0):8.2f} ms")
t0 = t(); _lib.check(lib.pope_copy_to_device(_lib.ptr(ei), _lib.ptr(out), nbytes, st)); torch.cuda.current_stream().synchronize(); print(f"the same copy again                   {ms(t0):8.2f} ms")
t0 = t(); lib.pope_host_unpin(_lib.ptr(ei)); print(f"hipHostUnregister                     {ms(t0):8.2f} ms")
t0 = t(); z = torch.zeros(1024, device=dev); torch.cuda.current_stream().synchronize(); print(f"first torch kernel (fill)             {ms(t0):8.2f} ms")
