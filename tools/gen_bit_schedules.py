"""Regenerate hawq_amd/bit_schedules.py's compact table from the reference's bit_config.py.

Run in the build container only (needs /root/reference).  The reference stores one
4.2 kLoC dict (bit_config.py:1-4204); we keep the same *data* as one short string per
ResNet schedule, in module order, and rebuild the dict at import time.
"""
import importlib.util
import sys

sys.path.insert(0, "/root/repo")
from hawq_amd.bit_schedules import module_names  # noqa: E402

spec = importlib.util.spec_from_file_location("refbc", "/root/reference/bit_config.py")
m = importlib.util.module_from_spec(spec)
spec.loader.exec_module(m)
code = {4: "4", 8: "8", 16: "g"}
for key, cfg in m.bit_config_dict.items():
    name = key[len("bit_config_"):]
    arch, scheme = name.split("_", 1)
    if not arch.startswith("resnet"):
        continue
    names = module_names(arch)
    assert list(cfg.keys()) == names, (key, [a for a, b in zip(cfg.keys(), names) if a != b][:5])
    print(f'    "{name}": "{"".join(code[cfg[n]] for n in names)}",')
