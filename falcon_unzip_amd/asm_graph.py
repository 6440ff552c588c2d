"""Reader of a FALCON assembly directory's graph files (`sg_edges_list`, `utg_data`, `ctg_paths`) -- the role of
`falcon_kit.fc_asm_graph.AsmGraph`, which `falcon_unzip/graphs_to_h_tigs.py:1,628-634` imports.  falcon_kit is not part of the
reference tree (un-vendored dependency, `setup.py:9`); this is a restatement of its published file formats:

    sg_edges_list   v w seq_id begin end score identity type          (one string-graph edge per line; type 'G' = kept)
    utg_data        s v t type length score path_or_edges             (simple: n0~n1~...; compound: s~v~t|s~v~t|...)
    ctg_paths       ctg_id ctg_type start_edge end_node length score s~v~t|s~v~t|...

Attributes used by the haplotig layout: `sg_edges[(v, w)] = ((seq_id, begin, end), score, identity, type)`,
`ctg_data[ctg_id] = (type, start_edge, end_node, length, score, ((s, v, t), ...))`, `get_sg_for_ctg(ctg_id)`.
"""
from __future__ import annotations

import networkx as nx


class AsmGraph:
    def __init__(self, sg_file, utg_file, ctg_file):
        self.sg_edges = {}
        self.utg_data = {}
        self.ctg_data = {}
        self.utg_to_ctg = {}
        with open(sg_file) as f:
            for line in f:
                t = line.split()
                if len(t) < 8:
                    continue
                self.sg_edges[(t[0], t[1])] = ((t[2], int(t[3]), int(t[4])), int(t[5]), float(t[6]), t[7])
        with open(utg_file) as f:
            for line in f:
                t = line.split()
                if len(t) < 7:
                    continue
                s, v, e = t[0:3]
                self.utg_data[(s, e, v)] = (t[3], int(t[4]), int(t[5]), t[6])
        with open(ctg_file) as f:
            for line in f:
                t = line.split()
                if len(t) < 7:
                    continue
                path = tuple(tuple(u.split("~")) for u in t[6].split("|"))
                self.ctg_data[t[0]] = (t[1], t[2], t[3], int(t[4]), int(t[5]), path)
                for s, v, e in path:
                    self.utg_to_ctg[(s, e, v)] = t[0]

    def _unitig_paths(self, kind, spec):
        if kind == "compound":
            for svt in spec.split("|"):
                s, v, e = svt.split("~")
                yield self.utg_data[(s, e, v)][3].split("~")
        else:
            yield spec.split("~")

    def get_sg_for_ctg(self, ctg_id):
        """the string-graph nodes and edges along the contig's unitigs, as a DiGraph"""
        g = nx.DiGraph()
        for s, v, e in self.ctg_data[ctg_id][5]:
            kind, _, _, spec = self.utg_data[(s, e, v)]
            for nodes in self._unitig_paths(kind, spec):
                nx.add_path(g, nodes)
        return g
