"""spectre_vit -- MI355X-native drop-in for the reference package of the same name.

Same import paths, class names, constructor arguments, forward signatures and state_dict keys as
Biblbrox/ViT-Spectre-Experiments' ``spectre_vit.models`` / ``spectre_vit.modules``; the
arithmetic runs in hand-written HIP kernels (libspv_hip.so, C-ABI in include/spv.h).
"""
