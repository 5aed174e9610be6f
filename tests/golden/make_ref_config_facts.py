#!/usr/bin/env python
"""Record, for every config file the reference ships (configs/*.json), the keys its engines consume -- one compact
fixture instead of copies of the files.  Build container only (reads /root/reference)."""
import json
import os

REF = "/root/reference/configs"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "ref_config_facts.json")
EXAMPLE_KEYS = ("generation_length", "max_turns", "template")           # consumed by the example scripts, not the engine

facts = {}
for name in sorted(os.listdir(REF)):
    with open(os.path.join(REF, name)) as f:
        cfg = json.load(f)
    facts[name] = {"engine_kwargs": {k: v for k, v in cfg.items() if k not in EXAMPLE_KEYS},
                   "example_keys": sorted(k for k in cfg if k in EXAMPLE_KEYS)}
with open(OUT, "w") as f:
    json.dump(facts, f, indent=1, sort_keys=True)
print(len(facts), "configs ->", OUT)
