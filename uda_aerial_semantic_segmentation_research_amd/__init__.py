"""MI355X-native (gfx950) hot path of bempt/uda_aerial_semantic_segmentation_research.

Host side mirrors the reference's model-creation / train-step API (``Unet``, ``DomainDiscriminator``,
``AdversarialLoss``, ``SegmentationTrainer``, ``AdversarialTrainer``); the arithmetic runs in hand-written HIP
kernels behind the C-ABI of ``include/udaseg.h`` (``libudaseg_hip.so``).  No CPU fallback.
"""
__version__ = "0.1.0"
