"""Config module 'vit_cifar100' (values: configs/_presets.py, which cites the reference file)."""
from spectre_vit.configs._presets import PRESETS as _P

globals().update(_P["vit_cifar100"])
