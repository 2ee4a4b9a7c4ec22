"""Hyper-parameter presets of the reference's config modules (values only; reference files cited per preset).

Each ``spectre_vit/configs/<name>.py`` of the reference is a flat list of module-level assignments; here the values live
in one table and the per-name modules materialise them, so ``parse_config("spectre_vit/configs/<name>.py")`` returns the
same attributes.  Quirk kept on purpose: only ``spectre_vit_cifar100`` spells the inheritance key
``__base__``; every other reference config spells it ``_base_``, which the parser drops, so those have no
``random_seed`` / ``learning_rate``.
"""

DEFAULT = dict(random_seed=42, learning_rate=1e-3)  # reference configs/default.py:1-2


def _common(img_size, embed_dim, num_heads, hidden_dim):
    patch_size = 4
    return dict(batch_size=8, val_batch_size=512, epochs=1000, num_classes=100, patch_size=patch_size, img_size=img_size,
                in_channels=3, num_heads=num_heads, dropout=0.001, hidden_dim=hidden_dim, adam_weight_decay=0.01,
                adam_betas=(0.9, 0.999), activation="gelu", num_encoders=4, embed_dim=embed_dim,
                num_patches=(img_size // patch_size) ** 2, use_spectre=True, spectre_threshold=1.0)


PRESETS = {
    # reference configs/spectre_vit_cifar100.py:1-22 (the one the scripts load; has __base__)
    "spectre_vit_cifar100": dict(_common(32, 512, 16, 768), __base__="default.py"),
    # reference configs/spectre_vit_mnist.py:3-19 (embed_dim = patch_size**2 * in_channels = 48)
    "spectre_vit_mnist": dict(_common(28, 48, 8, 256), _base_=["./default.py"]),
    # reference configs/vit_cifar100.py, fnet_cifar100.py: same Small/CIFAR values
    "vit_cifar100": dict(_common(32, 512, 16, 768), _base_=["./default.py"]),
    "fnet_cifar100": dict(_common(32, 512, 16, 768), _base_=["./default.py"]),
    # reference configs/spectre_branch.py:1-23: its own values (bs 512, 5000 epochs, 8 heads, hidden 256, embed 768), `_base_` a
    # plain string, and NO val_batch_size
    "spectre_branch": {k: v for k, v in dict(_common(32, 768, 8, 256), batch_size=512, epochs=5000, _base_="default.py").items()
                       if k != "val_batch_size"},
    # reference configs/vit_mnist.py, fnet_mnist.py: Small on 28 x 28
    "vit_mnist": dict(_common(28, 512, 16, 768), _base_=["./default.py"]),
    "fnet_mnist": dict(_common(28, 512, 16, 768), _base_=["./default.py"]),
}
