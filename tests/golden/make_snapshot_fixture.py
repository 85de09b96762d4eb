"""Generates tests/golden/snapshot_keys.json: the key paths the REFERENCE writes into a snapshot -- `Testbed::save_snapshot` (src/testbed.cu) assigns
m_network_config["snapshot"]... and snapshot[...]..., `Trainer::serialize` (tiny-cuda-nn/trainer.h) assigns data[...] -- parsed from the two functions' text.
/root/reference is read at generation time only.

Usage:  python tests/golden/make_snapshot_fixture.py
"""
import json
import os
import re
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from make_int_fixtures import REF, HERE, fragment  # noqa: E402


def main():
    save = fragment("src/testbed.cu", "void Testbed::save_snapshot(const std::string& filepath_string, bool include_optimizer_state) {")
    ser = fragment("dependencies/neus2_tcnn/include/tiny-cuda-nn/trainer.h", "json serialize(bool serialize_optimizer = false) {")
    paths = []
    for m in re.finditer(r'(?:m_network_config\["snapshot"\]|\bsnapshot)((?:\["\w+"\])+)\s*=', save):
        paths.append(["snapshot"] + re.findall(r'\["(\w+)"\]', m.group(1)))
    trainer = [["snapshot", k] for k in re.findall(r'data\["(\w+)"\]\s*=', ser)]
    assert ["snapshot", "training_step"] in paths and ["snapshot", "params_binary"] in trainer, (paths, trainer)
    out = {"source": "Testbed::save_snapshot (src/testbed.cu) and Trainer::serialize (tiny-cuda-nn/trainer.h) of RobinBruneau/RNb-NeuS2, parsed by tests/golden/make_snapshot_fixture.py",
           "written_by_save_snapshot": paths, "written_by_trainer_serialize": trainer,
           "optimizer_state_included_by_main": "include_optimizer_state" in save and False}
    with open(os.path.join(HERE, "snapshot_keys.json"), "w") as f:
        json.dump(out, f, indent=1)
    print(paths, trainer)


if __name__ == "__main__":
    main()
