"""Config module 'spectre_vit_mnist' (values: configs/_presets.py, which cites the reference file)."""
from spectre_vit.configs._presets import PRESETS as _P

globals().update(_P["spectre_vit_mnist"])
