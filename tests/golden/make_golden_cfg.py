"""Records the reference's YAML config surface (config/config.yaml, config/custom.yaml, config/category/*.yaml) as
parsed key/value DATA in tests/golden/category_configs.json, so a CPU test can check that this repo's config/ files
carry the same keys and values (camera.yaml / mug.yaml axis overrides included).

Run in the build container only:   python tests/golden/make_golden_cfg.py
"""
import glob
import json
import os

import yaml

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.environ.get("CPPF_GOLDEN_OUT", HERE)     # tests/test_golden_regen.py regenerates into a temp dir
REF = "/root/reference/config"


def load(p):
    with open(p) as f:
        return yaml.safe_load(f)


out = dict(config=load(os.path.join(REF, "config.yaml")), custom=load(os.path.join(REF, "custom.yaml")),
           category={os.path.basename(p)[:-5]: load(p) for p in sorted(glob.glob(os.path.join(REF, "category", "*.yaml")))})
with open(os.path.join(OUT, "category_configs.json"), "w") as f:
    json.dump(out, f, indent=1, sort_keys=True)
print(json.dumps(out["category"], indent=1))
