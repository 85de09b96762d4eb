"""Generates tests/golden/cli_flags.json from the REFERENCE's command line (src/main.cu): every `Flag` / `ValueFlag<T>` / `HelpFlag` / `Positional` it declares with the `args`
library -- kind, value type, placeholder, help text, short and long names -- parsed from the declarations' text. /root/reference is read at generation time only; the tests compare
`build/testbed -h` and the parser's behaviour with this list.

Usage:  python tests/golden/make_cli_fixture.py
"""
import json
import os
import re

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))


def main():
    src = open(os.path.join(REF, "src", "main.cu")).read()
    body = src[src.index("int main(int argc, char** argv)"):]
    flags = []
    for m in re.finditer(r"\b(HelpFlag|Flag|ValueFlag<\s*([\w:]+)\s*>|PositionalList<\s*([\w:]+)\s*>|Positional<\s*([\w:]+)\s*>)\s+(\w+)\s*\{\s*parser\s*,\s*\"([^\"]*)\"\s*,\s*\"((?:[^\"\\]|\\.)*)\"\s*(?:,\s*\{([^}]*)\})?", body):
        kind = m.group(1).split("<")[0]
        names = [t.strip() for t in (m.group(8) or "").split(",") if t.strip()]
        flags.append({"kind": kind, "value_type": m.group(2) or m.group(3) or m.group(4), "variable": m.group(5), "placeholder": m.group(6), "help": m.group(7),
                      "short": [t.strip("'") for t in names if t.startswith("'")], "long": [t.strip('"') for t in names if t.startswith('"')]})
    assert len(flags) >= 20 and any(f["long"] == ["no-gui"] for f in flags), len(flags)
    out = {"source": "src/main.cu of RobinBruneau/RNb-NeuS2 (the args declarations of main), parsed by tests/golden/make_cli_fixture.py", "flags": flags}
    with open(os.path.join(HERE, "cli_flags.json"), "w") as f:
        json.dump(out, f, indent=1)
    print(len(flags), "declarations:", [("--" + f["long"][0]) if f["long"] else f["placeholder"] for f in flags])


if __name__ == "__main__":
    main()
