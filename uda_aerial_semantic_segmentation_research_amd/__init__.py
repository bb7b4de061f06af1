"""MI355X-native (gfx950) hot path of bempt/uda_aerial_semantic_segmentation_research.

Host side mirrors the reference's model-creation / train-step API (``Unet``, ``DomainDiscriminator``,
``AdversarialLoss``, ``SegmentationTrainer``, ``AdversarialTrainer``); the arithmetic runs in hand-written HIP
kernels behind the C-ABI of ``include/udaseg.h`` (``libudaseg_hip.so``).  No CPU fallback.
"""
__version__ = "0.1.0"

import os as _os

# HIP maps streams onto a small pool of hardware queues (default 4).  This package uses three streams of its own (compute,
# weight-gradient side stream, all-reduce stream) and RCCL adds more; once two of them share a hardware queue their kernels
# serialise (measured: the side-stream overlap vanished as soon as a NCCL process group existed, -7 % step time; 8 queues
# restore it).  Read by the HIP runtime at its first call, so it must be set before any GPU work.
_os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
# Kernel arguments written straight to device memory instead of through a host-coherent staging buffer: shortens every launch on
# the host side.  BASELINE cfg 3 (~370 launches per iteration) runs close to the host's launch rate on the slower hosts of the pool:
# 7.89 -> 7.20 ms of host time per iteration there (tools/host_probe.py), cfg 2 unchanged (969.9 against 971.9 images/s).
_os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")
