"""Writes tests/golden/reference_config_keys.json: every leaf of the reference's configs/nerf/base.json as (key path, value) -- the schema a user of the reference
passes to `testbed --config`, from which tests rebuild a file of the same shape (the reference's file itself stays where it is; /root/reference is read here,
at generation time, only). Run in the build container:  python tests/golden/make_config_fixture.py"""
import json
import os

REF = "/root/reference/configs/nerf/base.json"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "reference_config_keys.json")


def leaves(node, path=()):
    if isinstance(node, dict):
        for k, v in node.items():
            yield from leaves(v, path + (k,))
    elif isinstance(node, list):
        for i, v in enumerate(node):
            yield from leaves(v, path + (i,))
    else:
        yield list(path), node


def main():
    with open(REF) as f:
        cfg = json.load(f)
    out = {"source": "configs/nerf/base.json of RobinBruneau/RNb-NeuS2 (key paths and values; integer path elements index arrays)", "leaves": list(leaves(cfg))}
    with open(OUT, "w") as f:
        json.dump(out, f, indent=0)
    print("%d leaves -> %s" % (len(out["leaves"]), OUT))


if __name__ == "__main__":
    main()
