"""a8: edge-list in / triplet out formats and the console entry point."""
import os

import numpy as np
import pytest
import scipy.sparse as sparse

from conftest import GOLDEN

from reveal_graph_embedding_amd.datautil.datarw import read_adjacency_matrix, write_features
from reveal_graph_embedding_amd.entry_points.arcte import build_parser, main


def test_reader_renumbers_in_first_seen_order_and_skips_comments(tmp_path):
    p = tmp_path / "e.tsv"
    p.write_text("# header\n10\t7\t1.5\n7\t3\t2\n3\t3\t4.0\n10\t3\t0.25\n")
    a, node_to_id = read_adjacency_matrix(str(p), "\t", undirected=False)
    assert node_to_id == {0: 10, 1: 7, 2: 3}
    assert a.shape == (3, 3)
    assert np.array_equal(a.toarray(), [[0, 1.5, 0.25], [0, 0, 2.0], [0, 0, 4.0]])
    a, _ = read_adjacency_matrix(str(p), "\t", undirected=True)
    assert np.array_equal(a.toarray(), [[0, 1.5, 0.25], [1.5, 0, 2.0], [0.25, 2.0, 4.0]])   # loop not doubled


def test_reader_matches_fixture_graph_shape():
    a, node_to_id = read_adjacency_matrix(os.path.join(GOLDEN, "cli_edges.tsv"), "\t", True)
    assert a.shape == (60, 60) and len(node_to_id) == 60
    assert (sparse.csr_matrix(a) != sparse.csr_matrix(a).T).nnz == 0


def test_writer_format(tmp_path):
    f = sparse.csr_matrix(np.array([[1.0, 0, 2.0], [0, 0, 0], [0, 3.0, 0]]))
    out = tmp_path / "f.tsv"
    write_features(str(out), f, ",", {0: 100, 1: 200, 2: 300})
    assert out.read_text() == "100,0,1\n100,2,2\n300,1,3\n"


def test_flags_and_defaults_match_the_reference():
    a = build_parser().parse_args(["-i", "in", "-o", "out"])
    assert (a.separator, a.undirected, a.restart_probability, a.epsilon_threshold, a.number_of_tasks) == \
        ("\t", False, 0.1, 1e-5, None)
    assert build_parser().parse_args(["-i", "a", "-o", "b", "-u", "False"]).undirected is True   # type=bool quirk
    b = build_parser().parse_args(["--input", "a", "--output", "b", "--separator", ",", "--rho", "0.2",
                                   "--epsilon", "1e-4", "--tasks", "3"])
    assert (b.separator, b.restart_probability, b.epsilon_threshold, b.number_of_tasks) == (",", 0.2, 1e-4, 3)


def test_formats_reproduce_reference_bytes_with_the_oracle_as_compute(tmp_path):
    """Reader -> (A + A^T)/2 -> [oracle arcte] -> writer gives the reference console script's exact bytes."""
    from oracle import oracle
    a, n2i = read_adjacency_matrix(os.path.join(GOLDEN, "cli_edges.tsv"), "\t", True)
    a = sparse.csr_matrix(a)
    a = (a + a.transpose()) / 2
    out = tmp_path / "features.tsv"
    write_features(str(out), sparse.csr_matrix(oracle.arcte(a, 0.1, 1e-5, 1)), "\t", n2i)
    assert out.read_text() == open(os.path.join(GOLDEN, "cli_features_expected.tsv")).read()


@pytest.mark.gpu
def test_console_script_reproduces_the_reference_output_bytes(tmp_path):
    out = tmp_path / "features.tsv"
    main(["-i", os.path.join(GOLDEN, "cli_edges.tsv"), "-o", str(out), "-u", "1", "-nt", "1"])
    got = out.read_text()
    want = open(os.path.join(GOLDEN, "cli_features_expected.tsv")).read()
    assert sorted(got.splitlines()) == sorted(want.splitlines())
    assert got == want


# ---- the native reader / writer (csrc/arcte_io.cpp) against a restatement of the reference's loops ------------------------

def _reference_reader(path, separator, undirected):
    """datarw.py:54-120 restated (test infrastructure): the loop the native reader replaces."""
    id_to_node, row, col, data = {}, [], [], []
    with open(path) as f:
        for line in f:
            words = line.strip().split(separator)
            if words[0][0] == "#":
                continue
            s = id_to_node.setdefault(int(words[0]), len(id_to_node))
            t = id_to_node.setdefault(int(words[1]), len(id_to_node))
            w = float(words[2])
            row.append(s); col.append(t); data.append(w)
            if undirected and s != t:
                row.append(t); col.append(s); data.append(w)
    ids = [None] * len(id_to_node)
    for k, v in id_to_node.items():
        ids[v] = k
    return len(id_to_node), np.array(row, np.int32), np.array(col, np.int32), np.array(data, np.float64), np.array(ids, np.int64)


@pytest.mark.parametrize("separator", ["\t", ",", " ", "::"])
@pytest.mark.parametrize("undirected", [False, True])
def test_native_reader_equals_the_reference_loop(tmp_path, separator, undirected, monkeypatch):
    from reveal_graph_embedding_amd.datautil.datarw import read_edge_triplets
    rng = np.random.default_rng(7)
    ids = rng.integers(-50, 10 ** 12, size=400)
    lines = ["# comment line", "#another%sone" % separator]
    for k in range(5000):
        a, b = ids[rng.integers(0, ids.size)], ids[rng.integers(0, ids.size)]
        w = ["1", "0.5", "2.25e-3", "-1.5", "3.", "1e5", "7.0" if separator == " " else " 7.0 "][k % 7]
        lines.append("%s%d%s%d%s%s%s" % ("  " if k % 11 == 0 else "", a, separator, b, separator, w, "\r" if k % 13 == 0 else ""))
        if k % 500 == 0:
            lines.append("# mid-file comment")
    lines.append("%d%s%d%s4.0%sextra field" % (ids[0], separator, ids[0], separator, separator))       # a self-loop, a fourth field
    p = tmp_path / "edges.txt"
    p.write_text("\n".join(lines))                                   # (no newline at the end of the file)
    want = _reference_reader(str(p), separator, undirected)
    for threads in ("1", "5"):
        monkeypatch.setenv("ARCTE_HIP_IO_THREADS", threads)
        got = read_edge_triplets(str(p), separator, undirected)
        assert got[0] == want[0]
        for g, w_ in zip(got[1:], want[1:]):
            assert g.dtype == w_.dtype and np.array_equal(g, w_)


def test_native_reader_follows_python_number_and_line_rules(tmp_path, monkeypatch):
    """What int() / float() / text-file iteration take and C's strtod / a split at "\\n" do not, and the other way round
    (round-3 advisor): single underscores between digits, lines that end in a lone "\\r", hexadecimal floats."""
    from reveal_graph_embedding_amd import _native
    from reveal_graph_embedding_amd.datautil.datarw import read_edge_triplets
    p = tmp_path / "edges.tsv"
    text = "1_0\t2_000\t1_0.5\r3\t1_0\t2e0\r\n2000\t3\t.5\r#c\r7\t7\t1"
    p.write_bytes(text.encode())
    want = _reference_reader(str(p), "\t", True)
    for threads in ("1", "3"):
        monkeypatch.setenv("ARCTE_HIP_IO_THREADS", threads)
        got = read_edge_triplets(str(p), "\t", True)
        assert got[0] == want[0] == 4
        for g, w_ in zip(got[1:], want[1:]):
            assert g.dtype == w_.dtype and np.array_equal(g, w_)
    for bad in ("1\t2\t0x10\n", "1\t2\t1__0\n", "1\t2\t_1\n", "1\t2\tnan(7)\n", "1_\t2\t1\n", "0x1\t2\t1\n"):
        p.write_text(bad)
        with pytest.raises((ValueError, IndexError)):
            _reference_reader(str(p), "\t", False)                   # the reference's own loop refuses it ...
        with pytest.raises(_native.ArcteHipError) as e:
            read_edge_triplets(str(p), "\t", False)                   # ... and so does the native reader
        assert e.value.code == -1, bad
    # many distinct ids: the id table starts small and grows
    rng = np.random.default_rng(11)
    ids = rng.permutation(200000)
    p.write_text("".join("%d\t%d\t1\n" % (ids[k], ids[(k * 7 + 1) % ids.size]) for k in range(100000)))
    want = _reference_reader(str(p), "\t", False)
    got = read_edge_triplets(str(p), "\t", False)
    assert got[0] == want[0]
    for g, w_ in zip(got[1:], want[1:]):
        assert np.array_equal(g, w_)


def test_native_reader_reports_the_bad_line(tmp_path):
    from reveal_graph_embedding_amd import _native
    from reveal_graph_embedding_amd.datautil.datarw import read_edge_triplets
    p = tmp_path / "bad.tsv"
    p.write_text("1\t2\t1.0\n3\tx\t1.0\n")
    with pytest.raises(_native.ArcteHipError) as e:
        read_edge_triplets(str(p), "\t", False)
    assert e.value.code == -1 and "line 2" in str(e.value)
    p.write_text("1\t2\t1.0\n\n4\t5\t1\n")                          # the reference raises IndexError on an empty line
    with pytest.raises(_native.ArcteHipError) as e:
        read_edge_triplets(str(p), "\t", False)
    assert "line 2" in str(e.value)
    with pytest.raises(_native.ArcteHipError):
        read_edge_triplets(str(tmp_path / "missing.tsv"), "\t", False)
    p.write_text("")
    n, row, col, val, ids = read_edge_triplets(str(p), "\t", True)
    assert n == 0 and row.size == 0 and ids.size == 0


def test_native_writer_equals_the_reference_loop(tmp_path, monkeypatch):
    from reveal_graph_embedding_amd.datautil.datarw import write_feature_triplets
    rng = np.random.default_rng(3)
    n = 700
    m = sparse.random(n, 2 * n, density=0.02, random_state=5, format="csr")
    m.data[:] = 1.0
    m = m + sparse.hstack([sparse.eye(n), sparse.csr_matrix((n, n))]).tocsr()          # every diagonal entry present
    m = sparse.csr_matrix(m)
    m.sort_indices()
    doubled = np.sort(rng.choice(n, size=40, replace=False))
    for i in doubled:
        m[i, i] = 2.0
    m.data[m.data > 2] = 1.0
    for i in range(n):                                                                     # diagonal of the others is 1
        if i not in set(doubled.tolist()):
            m[i, i] = 1.0
    m = sparse.csr_matrix(m)
    m.sort_indices()
    node_ids = rng.integers(-10, 10 ** 15, size=n)
    ref = tmp_path / "ref.tsv"
    write_features(str(ref), m, "\t", dict(enumerate(node_ids.tolist())))                  # the reference's per-entry loop
    for threads in ("1", "6"):
        monkeypatch.setenv("ARCTE_HIP_IO_THREADS", threads)
        out = tmp_path / ("native%s.tsv" % threads)
        write_feature_triplets(str(out), m.indptr, m.indices, doubled, "\t", node_ids)
        assert out.read_bytes() == ref.read_bytes()
    out = tmp_path / "dict.csv"
    write_feature_triplets(str(out), m.indptr, m.indices, doubled, ",", dict(enumerate(node_ids.tolist())))
    assert out.read_text() == ref.read_text().replace("\t", ",")
