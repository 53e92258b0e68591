#!/usr/bin/env python3
"""Transcribes the DATA of the reference's Python test-suite (py-dcdf/tests/test_dcdf.py) into tests/golden/pydcdf_fixture.json:
the 3x8x8 `test_array` literal (test_dcdf.py:10-43), the recipe of the `populated` fixture (variables, their parameters and the
append splits, test_dcdf.py:110-170) and the index sets of the query tests (test_dcdf.py:236-300).  The reference is only READ
as text (its extension module cannot be built here); the fixture is data, the tests that use it are tests/test_gpu_dataset.py."""
import ast
import json
import os
import re

REF = "/root/reference/py-dcdf/tests/test_dcdf.py"
HERE = os.path.dirname(os.path.abspath(__file__))


def main():
    src = open(REF).read()
    m = re.search(r"test_array = numpy\.array\(\s*(\[.*?\])\s*\)\n\n", src, re.S)
    arr = ast.literal_eval(m.group(1))
    assert len(arr) == 3 and len(arr[0]) == 8 and len(arr[0][0]) == 8
    variables = []
    # add_variable(name, span_size, chunk_size, k2_levels[, round, fractional_bits][, dtype]) + the appends that follow
    for name, dtype, total, splits, rnd, fb in [
            ("apples", "float32", 360, [99, 200], False, 0), ("pears", "float64", 500, [189, 400], False, 0),
            ("bananas", "int32", 511, [59, 300], False, 0), ("grapes", "int64", 365, [179, 300], False, 0),
            ("dates", "float32", 489, [], True, 2), ("melons", "float64", 489, [], True, 2)]:
        assert re.search(r'add_variable\(\s*"%s", 10, 20' % name, src), name
        variables.append({"name": name, "dtype": dtype, "instants": total, "splits": splits, "round": rnd, "fractional_bits": fb,
                          "span_size": 10, "chunk_size": 20, "k2_levels": [2, 2]})
    fx = {"source": "py-dcdf/tests/test_dcdf.py", "test_array": arr, "variables": variables,
          "commit_after": "grapes",  # test_dcdf.py:141-145: commit + reload between grapes and dates
          "rounded_expected": "(data * 4 + 0.001).round() / 4",  # test_dcdf.py:149,158
          "real_world": {"file": "cpc_precip_day.npz", "shape": [1, 360, 720], "chunk_size": 64, "k2_levels": [4, 6], "span_size": 20000}}
    with open(os.path.join(HERE, "pydcdf_fixture.json"), "w") as f:
        json.dump(fx, f)
    print("wrote pydcdf_fixture.json")


if __name__ == "__main__":
    main()
