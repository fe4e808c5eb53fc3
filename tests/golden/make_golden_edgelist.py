#!/usr/bin/env python3
"""Golden vectors for the edge-list reader: the reference's own AdjlstGraph (src/pecanpy/graph.py:108-386)
is run on a set of small edge-list texts; inputs (the texts) and outputs (IDs, CSR, dense matrix,
num_edges, warning count or exception type) are stored as data in tests/golden/edgelist_cases.json.
Runs only in the build container (needs /root/reference and the stub packages of make_golden.py).

usage:  python tests/golden/make_golden_edgelist.py
"""
import json
import os
import sys
import tempfile
import warnings

HERE = os.path.dirname(os.path.abspath(__file__))
sys.dont_write_bytecode = True
sys.path.insert(0, os.path.join(HERE, "_stubs"))
sys.path.insert(1, "/root/reference/src")

import numpy as np  # noqa: E402
from pecanpy.graph import AdjlstGraph  # noqa: E402  (the reference)


def cases():
    rng = np.random.default_rng(7)
    out = []

    def add(name, text, weighted, directed, delimiter="\t"):
        out.append(dict(name=name, text=text, weighted=weighted, directed=directed, delimiter=delimiter))

    add("plain_unweighted", "a\tb\nb\tc\nc\ta\nd\ta\n", False, False)
    add("plain_directed", "a\tb\nb\tc\nc\ta\nd\ta\n", False, True)
    add("no_trailing_newline", "x\ty\ny\tz", False, False)
    add("crlf", "x\ty\r\ny\tz\r\n", False, False)
    add("extra_columns_unweighted", "a\tb\t0.5\tjunk\nb\tc\t7\n", False, False)
    add("spaces_around_ids", " a \t b \nb\t c\n", False, False)
    add("empty_id", "a\t\tb\n", False, False)
    add("self_loop_undirected", "a\ta\na\tb\n", False, False)
    add("duplicate_same_weight", "a\tb\t2\nb\ta\t2\na\tb\t2.0\n", True, False)
    add("duplicate_conflict", "a\tb\t1\nb\ta\t2\n", True, False)
    add("weights_formats", "a\tb\t1e-3\nb\tc\t.5\nc\td\t5.\nd\te\t+2\ne\tf\t 3.25 \n", True, False)
    add("weights_tenth", "a\tb\t0.1\nb\tc\t0.7\na\tc\t123456.789\n", True, True)
    add("nonpositive_weight", "a\tb\t0\nb\tc\t-1\nc\td\t2\n", True, False)
    add("nan_weight", "a\tb\tnan\nb\tc\t1\n", True, False)
    add("underscore_weight", "a\tb\t1_0\n", True, False)
    add("comma_delimiter", "a,b,1.5\nb,c,2.5\n", True, False, ",")
    add("multichar_delimiter", "a::b::1.5\nb::c::2.5\n", True, True, "::")
    add("space_delimiter", "a b\nb c\n", False, False, " ")
    # numeric-looking ids are strings: "7", "07", "+7", "7.0" are four different vertices
    add("numeric_id_spellings", "7\t07\n07\t+7\n+7\t7.0\n7.0\t7\n0\t00\n99999999\t100000000\n"
                                "67108863\t67108864\n123456789012\t7\n-1\t0\n", False, False)
    add("numeric_ids_out_of_order", "".join(f"{a}\t{b}\n" for a, b in [(5, 3), (3, 9), (9, 5), (0, 5), (12, 0), (3, 5)]),
        False, True)
    add("wrong_columns_weighted", "a\tb\n", True, False)
    add("blank_line", "a\tb\n\nb\tc\n", False, False)
    add("single_column", "abc\n", False, False)
    # random multigraph with repeats, self loops and a few isolated-by-direction vertices
    n, m = 60, 400
    src, dst = rng.integers(0, n, m), rng.integers(0, n, m)
    add("random_unweighted", "".join(f"v{s}\tv{d}\n" for s, d in zip(src, dst)), False, False)
    add("random_directed", "".join(f"v{s}\tv{d}\n" for s, d in zip(src, dst)), False, True)
    w = np.round(rng.random(m) * 9 + 0.5, 3)
    sym = {}
    lines = []
    for s, d, x in zip(src, dst, w):   # consistent weights per undirected pair -> no conflicts
        key = (min(s, d), max(s, d))
        x = sym.setdefault(key, x)
        lines.append(f"v{s}\tv{d}\t{x}\n")
    add("random_weighted", "".join(lines), True, False)
    return out


def run_reference(case):
    res = dict(case)
    with tempfile.NamedTemporaryFile("w", suffix=".edg", delete=False, newline="") as f:
        f.write(case["text"])
        path = f.name
    try:
        g = AdjlstGraph()
        with warnings.catch_warnings(record=True) as caught:
            warnings.simplefilter("always")
            try:
                g.read(path, case["weighted"], case["directed"], case["delimiter"])
            except Exception as exc:  # noqa: BLE001 - the exception type is the expected output
                res["error"] = type(exc).__name__
                return res
        indptr, indices, data = g.to_csr()
        res.update(error="", n_warnings=len(caught), ids=list(g.nodes), num_edges=int(g.num_edges),
                   indptr=indptr.tolist(), indices=indices.tolist(),
                   data_bits=data.view(np.uint32).tolist(),
                   dense_bits=g.to_dense().view(np.uint64).ravel().tolist(),
                   edges=[[int(h), int(t), float(w).hex()] for h, t, w in g.edges])
    finally:
        os.unlink(path)
    return res


def main():
    results = [run_reference(c) for c in cases()]
    with open(os.path.join(HERE, "edgelist_cases.json"), "w") as f:
        json.dump(results, f, separators=(",", ":"))
    for r in results:
        print(f"{r['name']:28s} error={r['error'] or '-':12s} warnings={r.get('n_warnings', '-')}")


if __name__ == "__main__":
    main()
