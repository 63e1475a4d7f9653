"""ORACLE -- TEST INFRASTRUCTURE ONLY.

CPU restatements of the Obj-GAN image_generation hot path used as the parity checker of the
gfx950 kernels.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this package; the product path (obj-gan_amd/) never does and fails loudly without its HIP library.

  roi.py         ctypes bindings of oracle/liboracle_roi.so (our C restatement) and, when built,
                 oracle/_ref/libref_roi_align.so (the reference's own roi_align.c)
  torch_ref.py   plain-PyTorch (CPU, fp32) restatement of attention / masked max / blocks / losses
  ref_harness.py imports the UNMODIFIED reference from /root/reference (only in the build
                 container) to validate the restatements and generate tests/golden/*
"""
