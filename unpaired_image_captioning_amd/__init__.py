"""MI355X-native hot path of gujiuxiang/unpaired_image_captioning (TopDown captioner training /
decoding step) behind the reference's own Python interface: `models.setup(opt)`,
`model(fc, attri, att, seq, att_masks)`, `model(..., mode='sample')`, `misc.criterion.*`.
All arithmetic runs in libuic_hip.so (hand-written gfx950 HIP kernels, C ABI in include/uic_hip.h)."""
__version__ = "0.1.0"
