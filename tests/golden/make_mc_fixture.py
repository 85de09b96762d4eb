"""Writes tests/golden/mc_triangle_table.json: the 256-case marching-cubes triangle table the reference triangulates with
(src/marching_cubes.cu:401-659 -- the public Bourke / PyMCubes `triangle_table`: constant data, not code), as 256 lists of edge ids.
Run in the build container, where /root/reference exists; the tests read only the JSON.

  python tests/golden/make_mc_fixture.py
"""
import json
import os
import re

REF = "/root/reference/src/marching_cubes.cu"
HERE = os.path.dirname(os.path.abspath(__file__))


def main():
    src = open(REF).read()
    i = src.index("triangle_table[256][16]")
    body = src[src.index("=", i) + 1:src.index("};", i)]
    nums = [int(x) for x in re.findall(r"-?\d+", body)]
    assert len(nums) == 256 * 16, len(nums)
    table = []
    for m in range(256):
        row = nums[m * 16:(m + 1) * 16]
        n = row.index(-1)
        assert n % 3 == 0 and all(v == -1 for v in row[n:]) and all(0 <= v < 12 for v in row[:n])
        table.append(row[:n])
    with open(os.path.join(HERE, "mc_triangle_table.json"), "w") as f:
        json.dump({"source": "src/marching_cubes.cu:401-659 (triangle_table; Bourke polygonise / PyMCubes, BSD-3-Clause)",
                   "corner_bits": "bit c of the case index = lattice corner c inside (density > threshold); corners (0,0,0) (1,0,0) (1,1,0) (0,1,0) (0,0,1) (1,0,1) (1,1,1) (0,1,1)",
                   "edges": "0-3: bottom face 0-1 1-2 2-3 3-0; 4-7: top face 4-5 5-6 6-7 7-4; 8-11: verticals 0-4 1-5 2-6 3-7",
                   "triangles": table}, f)
    print("wrote", len(table), "cases,", sum(len(t) for t in table) // 3, "triangles")


if __name__ == "__main__":
    main()
