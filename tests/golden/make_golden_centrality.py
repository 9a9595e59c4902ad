#!/usr/bin/env python3
"""Golden fixtures for the reference's OTHER driver: arcte_and_centrality (embedding/arcte/cython_opt/arcte.pyx:125-241)
and the feature weighting it ends with (embedding/common.py:8-67), by RUNNING THE REFERENCE.

Runs only in the build container (reference mounted read-only at /root/reference):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_centrality.py

The function lives in a Cython module the reference never builds (setup.py has every ext_module commented out), so
this recipe compiles the four .pyx files FROM WHERE THEY LIE with Cython + gcc into a temporary directory (nothing
is written to /root/reference, nothing of its source is stored here) and imports them through a throw-away package
skeleton whose __path__ points back at the reference for everything that is plain Python.

One .npz per graph:
  adj_*                 the adjacency matrix handed to the reference (the inputs are the graphs of make_golden.py
                        with isolated nodes removed: the reference raises on a node without out-edges once any
                        seed has run, arcte.pyx:210)
  rho, epsilon          raw epsilon: this driver has no effective-epsilon rule (arcte.pyx:172-180)
  feat_*                returned feature matrix (CSR, after normalize_community_features)
  centrality            returned centrality vector (the reference returns a 1 x n np.matrix; stored flat)
  ambiguous             per seed: does a node OUTSIDE the closed neighbourhood tie with the smallest value inside it?
                        Then the community depends on the order numpy's unstable argsort leaves ties in
                        (arcte.pyx:194-208) and the reference's answer is one of several legal ones.
  ncols_local           number of local-community columns
Also: weighting_*.npz for embedding/common.py normalize_columns / normalize_rows and
embedding/community_weighting.py on the n x 2n output of arcte() (fixtures of make_golden.py) with seeded labels.
"""
import os
import subprocess
import sys
import sysconfig
import tempfile

import numpy as np
import scipy.sparse as sparse

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, os.path.dirname(HERE))

PYX = [
    "reveal_graph_embedding/eps_randomwalk/cython_opt/transition.pyx",
    "reveal_graph_embedding/eps_randomwalk/cython_opt/push.pyx",
    "reveal_graph_embedding/eps_randomwalk/cython_opt/similarity.pyx",
    "reveal_graph_embedding/embedding/arcte/cython_opt/arcte.pyx",
]


def build_reference_extensions(tmp):
    """cythonize + gcc, outputs under `tmp` only."""
    inc = [sysconfig.get_paths()["include"], np.get_include()]
    suffix = sysconfig.get_config_var("EXT_SUFFIX")
    for rel in PYX:
        src = os.path.join(REF, rel)
        mod = rel[:-4].replace("/", ".")
        c_file = os.path.join(tmp, "build", mod + ".c")
        os.makedirs(os.path.dirname(c_file), exist_ok=True)
        subprocess.run([sys.executable, "-m", "cython", "-3", "--module-name", mod, "-o", c_file, src], check=True,
                       capture_output=True)
        out = os.path.join(tmp, "pkg", rel[:-4] + suffix)
        os.makedirs(os.path.dirname(out), exist_ok=True)
        subprocess.run(["gcc", "-O2", "-fPIC", "-shared", "-fwrapv", "-fno-strict-aliasing", "-w"] +
                       ["-I" + i for i in inc] + ["-o", out, c_file], check=True, capture_output=True)
    # package skeleton: every package directory of the compiled modules, chained to the reference's own directory
    for rel in PYX:
        d = os.path.dirname(rel)
        while d:
            init = os.path.join(tmp, "pkg", d, "__init__.py")
            if not os.path.exists(init):
                with open(init, "w") as f:
                    f.write("__path__.append(%r)\n" % os.path.join(REF, d))
            d = os.path.dirname(d)
    sys.path.insert(0, os.path.join(tmp, "pkg"))


def drop_isolated(a):
    a = sparse.csr_matrix(a)
    keep = np.where((np.diff(a.indptr) > 0) | (np.asarray(a.sum(axis=0)).reshape(-1) != 0))[0]
    return sparse.csr_matrix(a[keep][:, keep])


def graphs():
    from conftest import load_golden
    out = {}
    for name in ("ba300", "weighted", "selfloop", "grid25", "ws1000", "rmat2000", "directed"):
        a = drop_isolated(load_golden(name)["adjacency"])
        if name == "directed":
            # keep only nodes with out-edges AND in-edges inside the kept set (iterate: removing one may strand another)
            while True:
                ok = (np.diff(a.indptr) > 0) & (np.asarray(a.sum(axis=0)).reshape(-1) != 0)
                if ok.all():
                    break
                keep = np.where(ok)[0]
                a = sparse.csr_matrix(a[keep][:, keep])
        out[name] = a
    return out


def ambiguity(a, rho, epsilon):
    """Per seed: a support node outside the closed neighbourhood whose normalised value EQUALS the smallest
    normalised value inside it (the scan of arcte.pyx:200-208 may or may not take it)."""
    from reveal_graph_embedding.eps_randomwalk.transition import get_natural_random_walk_matrix
    from reveal_graph_embedding.eps_randomwalk.similarity import fast_approximate_cumulative_pagerank_difference
    w, od, idg = get_natural_random_walk_matrix(a)
    n = a.shape[0]
    a_i = np.ndarray(n, dtype=np.ndarray)
    w_i = np.ndarray(n, dtype=np.ndarray)
    for i in range(n):
        a_i[i] = w.indices[w.indptr[i]:w.indptr[i + 1]].astype(np.int64)
        w_i[i] = w.data[w.indptr[i]:w.indptr[i + 1]]
    amb = np.zeros(n, dtype=np.uint8)
    for seed in range(n):
        s = np.zeros(n)
        r = np.zeros(n)
        fast_approximate_cumulative_pagerank_difference(s, r, w_i, a_i, od, idg, seed, rho, epsilon)
        nz = np.nonzero(s)[0]
        sn = s[nz] / idg[nz]
        base = np.zeros(n, dtype=bool)
        base[a_i[seed]] = True
        base[seed] = True
        inb = base[nz]
        if inb.sum() < base.sum():
            continue                    # not every base member in the support: nothing is emitted
        thr = sn[inb].min()
        amb[seed] = np.any(sn[~inb] == thr)
    return amb


def main():
    tmp = tempfile.mkdtemp(prefix="refcy_")
    build_reference_extensions(tmp)
    from reveal_graph_embedding.embedding.arcte.cython_opt.arcte import arcte_and_centrality
    from reveal_graph_embedding.embedding.common import normalize_columns, normalize_rows
    from reveal_graph_embedding.embedding.community_weighting import (chi2_contingency_matrix, community_weighting,
                                                                      peak_snr_weight_aggregation)

    rho, epsilon = 0.1, 1e-4
    for name, a in graphs().items():
        # (handed over as COO, the documented input type; the reference then normalises ITS OWN csr copy in place --
        #  cython_opt/transition.pyx:19 has no copy=True -- so the base block it stacks is I + W, not I + A)
        f, c = arcte_and_centrality(sparse.coo_matrix(a, copy=True), rho, epsilon)
        f = sparse.csr_matrix(f)
        f.sort_indices()
        amb = ambiguity(a.copy(), rho, epsilon)
        n = a.shape[0]
        np.savez_compressed(os.path.join(HERE, "centrality_%s.npz" % name),
                            n=n, rho=rho, epsilon=epsilon,
                            adj_indptr=a.indptr.astype(np.int64), adj_indices=a.indices.astype(np.int64), adj_data=a.data,
                            feat_indptr=f.indptr.astype(np.int64), feat_indices=f.indices.astype(np.int64), feat_data=f.data,
                            feat_shape=np.array(f.shape, dtype=np.int64),
                            centrality=np.asarray(c, dtype=np.float64).reshape(-1), ambiguous=amb,
                            ncols_local=f.shape[1] - n)
        print("%-10s n=%5d nnz=%7d features %s nnz %d, local columns %d, ambiguous seeds %d" % (
            name, n, a.nnz, f.shape, f.nnz, f.shape[1] - n, int(amb.sum())), flush=True)

    # feature weighting on arcte()'s own output (the n x 2n matrices already pinned by make_golden.py)
    from conftest import load_golden
    for name in ("ba300", "weighted", "rmat2000", "selfloop"):
        g = load_golden(name)
        x = g["feat1"]
        n = g["n"]
        rng = np.random.default_rng(7)
        y = rng.integers(0, 5, size=n)
        train = np.sort(rng.choice(n, size=n // 2, replace=False))
        test = np.setdiff1d(np.arange(n), train)
        nc = sparse.csr_matrix(normalize_columns(x.copy()))
        nr = sparse.csr_matrix(normalize_rows(nc.copy()))
        x_train, x_test = sparse.csr_matrix(nc[train]), sparse.csr_matrix(nc[test])
        cm = chi2_contingency_matrix(x_train, y[train])
        wts = peak_snr_weight_aggregation(cm.copy())
        xt, xs = community_weighting(x_train.copy(), x_test.copy(), wts)
        xt, xs = sparse.csr_matrix(xt), sparse.csr_matrix(xs)
        for m in (nc, nr, xt, xs):
            m.sort_indices()
        np.savez_compressed(os.path.join(HERE, "weighting_%s.npz" % name), n=n, labels=y, train=train, test=test,
                            nc_indptr=nc.indptr, nc_indices=nc.indices, nc_data=nc.data,
                            nr_indptr=nr.indptr, nr_indices=nr.indices, nr_data=nr.data,
                            contingency=cm, weights=wts,
                            xt_indptr=xt.indptr, xt_indices=xt.indices, xt_data=xt.data,
                            xs_indptr=xs.indptr, xs_indices=xs.indices, xs_data=xs.data)
        print("weighting %-10s train %d test %d  nnz nc %d xt %d xs %d" % (name, train.size, test.size, nc.nnz, xt.nnz, xs.nnz),
              flush=True)


if __name__ == "__main__":
    main()
