#!/usr/bin/env python3
"""Golden vectors for the haplotig layout: RUN the reference's falcon_unzip/graphs_to_h_tigs.py (from /root/reference, in this container
only) on synthetic assembly graphs and store inputs + outputs as data under tests/golden_htigs/<case>/.

The reference is Python 2 + networkx 1.x + falcon_kit; here it is translated in memory (lib2to3), and given stand-ins for what is absent:
  falcon_kit.fc_asm_graph.AsmGraph   a reader of sg_edges_list / utg_data / ctg_paths (falcon_kit's published formats; not in the reference tree)
  falcon_kit.FastaReader.FastaReader  records with .name / .sequence
  networkx                           a 1.x face over the installed 3.x: DiGraph whose nodes()/edges()/in_edges()/out_edges() return lists and
                                     that has .node / .edge / add_path, weakly_connected_component_subgraphs; write_gexf is a no-op
  multiprocessing.Pool               map() in-process
Iteration orders that Python 2 leaves to dict hashing (graphs_to_h_tigs.py:98,150,430,537) are Python 3's insertion orders here -- the same
canonicalisation the other goldens of this repo use for py2-dict-ordered outputs.  Nothing of the reference is stored: only data.
usage: make_golden_htigs.py [case ...]"""
import gzip
import hashlib
import io
import json
import os
import shutil
import sys
import tempfile
import types

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, REPO)
REF = "/root/reference/falcon_unzip/graphs_to_h_tigs.py"

HOM, HET, TIE = "hom", "het", "tie"
CASES = {
    "h1_two_bubbles": (11, [("000000F", [(HOM, 20000), (HET, 30000), (HOM, 15000), (HET, 25000), (HOM, 20000)])]),
    "h2_three_contigs": (12, [("000000F", [(HOM, 18000), (HET, 40000), (HOM, 18000)]), ("000001F", [(HOM, 30000)]),
                              ("000002F", [(HOM, 15000), (HET, 22000), (HOM, 12000), (HET, 21000), (HOM, 12000), (HET, 30000), (HOM, 15000)])]),
    "h3_short_bubble": (13, [("000000F", [(HOM, 25000), (HET, 9000), (HOM, 25000)])]),      # alternative path of <= 5 edges: no haplotig (:467)
    # a bubble whose reads were not phased and whose two branches have the same number of edges: every edge scores 1, the source-to-sink routes through
    # either branch cost the same (graphs_to_h_tigs.py:354), and so do the candidates of the haplotig peeling (:505).  Which one the reference takes is
    # left to networkx (1.x under Python 2: heap ties by node comparison over hash-ordered adjacency -- unspecified; here, 3.x: first discovered = edge
    # insertion order = the primary assembly's edges first).  The mirror pins: least score, then fewest edges, then smallest predecessor name walking
    # back from the target -- on this case the same route (the primary assembly's reads carry the smaller ids).
    "h4_tied_bubble": (14, [("000000F", [(HOM, 20000), (HET, 30000), (HOM, 15000), (TIE, 18000), (HOM, 20000)])]),
}
OUTPUTS = ("p_ctg.%s.fa", "p_ctg_path.%s", "p_ctg_edges.%s", "h_ctg_all.%s.fa", "h_ctg_path.%s", "h_ctg_edges.%s", "path_len.%s")


def nx1_module():
    import networkx as nx

    from networkx.classes.reportviews import InDegreeView, InEdgeView, NodeView, OutDegreeView, OutEdgeView

    class _Listing:
        """G.nodes() / G.edges() / G.in_edges(n) / G.out_edges(n) as lists (networkx 1.x); G.nodes[...] still works"""
        def __init__(self, view):
            self.view = view

        def __call__(self, *a, **k):
            return list(self.view(*a, **k))

        def __getitem__(self, k):
            return self.view[k]

        def __iter__(self):
            return iter(list(self.view))

        def __len__(self):
            return len(self.view)

        def __contains__(self, x):
            return x in self.view

        def __getattr__(self, name):          # .items(), .data(), ...: what networkx itself uses
            return getattr(self.view, name)

    class DiGraph(nx.DiGraph):
        nodes = property(lambda self: _Listing(NodeView(self)))
        edges = property(lambda self: _Listing(OutEdgeView(self)))
        out_edges = property(lambda self: _Listing(OutEdgeView(self)))
        in_edges = property(lambda self: _Listing(InEdgeView(self)))
        in_degree = property(lambda self: InDegreeView(self))
        out_degree = property(lambda self: OutDegreeView(self))

        @property
        def node(self):
            return self._node

        @property
        def edge(self):
            return self._adj

        def add_path(self, nodes, **attr):
            nx.add_path(self, nodes, **attr)

    m = types.ModuleType("networkx")
    m.DiGraph = DiGraph
    m.shortest_path = nx.shortest_path
    m.shortest_path_length = lambda g, source=None, **k: dict(nx.shortest_path_length(g, source=source, **k))
    m.descendants = nx.descendants
    m.weakly_connected_component_subgraphs = lambda g: [g.subgraph(c).copy() for c in nx.weakly_connected_components(g)]
    m.write_gexf = lambda *a, **k: None
    m.exception = nx.exception
    return m


def load_reference(nx1):
    from lib2to3 import refactor
    with open(REF) as f:
        src = f.read()
    tool = refactor.RefactoringTool(refactor.get_fixers_from_package("lib2to3.fixes"))
    py3 = str(tool.refactor_string(src + "\n", REF))

    class AsmGraph:
        def __init__(self, sg_file, utg_file, ctg_file):
            self.sg_edges, self.utg_data, self.ctg_data = {}, {}, {}
            for l in open(sg_file):
                l = l.strip().split()
                self.sg_edges[(l[0], l[1])] = ((l[2], int(l[3]), int(l[4])), int(l[5]), float(l[6]), l[7])
            for l in open(utg_file):
                l = l.strip().split()
                self.utg_data[(l[0], l[2], l[1])] = (l[3], int(l[4]), int(l[5]), l[6])
            for l in open(ctg_file):
                l = l.strip().split()
                self.ctg_data[l[0]] = (l[1], l[2], l[3], int(l[4]), int(l[5]), tuple(e.split("~") for e in l[6].split("|")))

        def get_sg_for_ctg(self, ctg_id):
            sg = nx1.DiGraph()
            for s, v, t in self.ctg_data[ctg_id][-1]:
                type_, length, score, path_or_edges = self.utg_data[(s, t, v)]
                if type_ == "simple":
                    sg.add_path(path_or_edges.split("~"))
                else:
                    for svt in path_or_edges.split("|"):
                        s2, v2, t2 = svt.split("~")
                        sg.add_path(self.utg_data[(s2, t2, v2)][3].split("~"))
            return sg

    class Rec:
        def __init__(self, name, seq):
            self.name, self.sequence = name, seq

    def FastaReader(fn):
        name, chunks = None, []
        for line in open(fn):
            line = line.rstrip("\n")
            if line.startswith(">"):
                if name is not None:
                    yield Rec(name, "".join(chunks))
                name, chunks = line[1:].split()[0], []
            else:
                chunks.append(line)
        if name is not None:
            yield Rec(name, "".join(chunks))

    class Pool:
        def __init__(self, n):
            pass

        def map(self, f, xs):
            return [f(x) for x in xs]
    fk = types.ModuleType("falcon_kit")
    fag = types.ModuleType("falcon_kit.fc_asm_graph"); fag.AsmGraph = AsmGraph
    ffr = types.ModuleType("falcon_kit.FastaReader"); ffr.FastaReader = FastaReader
    mpm = types.ModuleType("multiprocessing"); mpm.Pool = Pool
    saved = {k: sys.modules.get(k) for k in ("falcon_kit", "falcon_kit.fc_asm_graph", "falcon_kit.FastaReader", "networkx", "multiprocessing")}
    sys.modules.update({"falcon_kit": fk, "falcon_kit.fc_asm_graph": fag, "falcon_kit.FastaReader": ffr, "networkx": nx1, "multiprocessing": mpm})
    mod = types.ModuleType("ref_graphs_to_h_tigs")
    try:
        exec(compile(py3, REF, "exec"), mod.__dict__)
    finally:
        for k, v in saved.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v
    return mod


def run_case(name, mod):
    from falcon_unzip_amd import sim_asm
    seed, layouts = CASES[name]
    work = tempfile.mkdtemp(prefix="htigs_")
    sim_asm.make_case(work, seed, layouts)
    cwd = os.getcwd()
    os.chdir(work)
    try:
        mod.main(["fc_graphs_to_h_tigs.py", "--fc_asm_path", "2-asm-falcon", "--fc_hasm_path", "1-hasm", "--ctg_id", "all", "--rid_phase_map", "rid_to_phase.all",
                  "--fasta", "preads4falcon.fasta"])
    finally:
        os.chdir(cwd)
    out = os.path.join(HERE, name)
    shutil.rmtree(out, ignore_errors=True)
    os.makedirs(out)
    manifest = {"case": name, "seed": seed, "layouts": layouts, "outputs": {}}
    for ctg, _ in layouts:
        d = os.path.join(work, ctg)
        if not os.path.isdir(d):
            manifest["outputs"][ctg] = None                    # contig skipped by the layout (no rows in the phase map)
            continue
        manifest["outputs"][ctg] = {}
        for pat in OUTPUTS:
            fn = pat % ctg
            with open(os.path.join(d, fn), "rb") as f:
                data = f.read()
            manifest["outputs"][ctg][fn] = {"bytes": len(data), "sha256": hashlib.sha256(data).hexdigest()}
            with gzip.GzipFile(os.path.join(out, "%s.%s.gz" % (ctg, fn)), "wb", mtime=0) as g:
                g.write(data)
    with open(os.path.join(out, "manifest.json"), "w") as f:
        json.dump(manifest, f, indent=1, sort_keys=True)
    shutil.rmtree(work, ignore_errors=True)
    return manifest


if __name__ == "__main__":
    nx1 = nx1_module()
    mod = load_reference(nx1)
    for name in (sys.argv[1:] or sorted(CASES)):
        m = run_case(name, mod)
        print(name, {c: (None if v is None else {k: x["bytes"] for k, x in v.items()}) for c, v in m["outputs"].items()})
