"""Haplotig layout: mirror of `falcon_unzip/graphs_to_h_tigs.py` (the consumer of `rid_to_phase.all`) for
Python 3 / networkx >= 2 -- same CLI flags, same files, same graph surgery in the same order.

What it does per primary contig (reference lines in brackets): builds a phase-annotated string graph from the contig's
primary edges ("OP") and the phased assembly's extra edges ("H", plus "ext" hooks at dead ends) with every edge mirrored
on the reverse strand [53-201]; drops haplotype components that touch both strands or none [203-258]; scores edges by
phase agreement [263-281]; removes short cuts that repeats create [314-347]; the cheapest source-to-sink path becomes the
updated primary contig [352-405]; what is left is peeled, longest path first, into haplotigs [409-560].  Outputs in
`./<ctg_id>/`: p_ctg.<ctg>.fa, p_ctg_path.<ctg>, p_ctg_edges.<ctg>, h_ctg_all.<ctg>.fa, h_ctg_path.<ctg>,
h_ctg_edges.<ctg>, path_len.<ctg>, sg.gexf, sg2.gexf.

This is CPU work by nature (pointer chasing over a few thousand nodes per contig): no kernels here; the GPU side of row n3
is the pile consensus (K6).  Iteration orders the reference leaves to Python-2 dict hashing (its lines 98, 150, 430, 537)
are insertion orders here; tests/golden_htigs/ pins the outputs against the reference run the same way
(tests/golden_htigs/make_golden_htigs.py).  `falcon_kit`'s AsmGraph / FastaReader are replaced by asm_graph.AsmGraph and
a local FASTA reader."""
from __future__ import annotations

import argparse
import os
import sys

import networkx as nx

from .asm_graph import AsmGraph

_RC = dict(zip("ACGTacgtNn-", "TGCAtgcaNn-"))


def reverse_end(node_id):
    rid, end = node_id.split(":")
    return rid + (":B" if end == "E" else ":E")


def _mirror(v, w):
    return reverse_end(w), reverse_end(v)


def _tag(ph):
    return "%d_%d" % ph


def _edge_seq(seqs, edge_data):
    seq_id, s, t = edge_data[0]
    if s < t:
        return seqs[seq_id][s:t]
    return "".join(_RC[c] for c in seqs[seq_id][s:t:-1])


def load_sg_seq(all_read_ids, fasta_fn):
    """p-read sequences of the reads that matter, upper-cased [28-36]"""
    seqs, name, chunks = {}, None, []

    def flush():
        if name is not None and name in all_read_ids:
            seqs[name] = "".join(chunks).upper()
    with open(fasta_fn) as f:
        for line in f:
            line = line.rstrip("\r\n")
            if line.startswith(">"):
                flush()
                name, chunks = (line[1:].split() or [""])[0], []
            else:
                chunks.append(line)
    flush()
    return seqs


def _drop_isolated(g):
    for v in list(g.nodes()):
        if g.out_degree(v) == 0 and g.in_degree(v) == 0:
            g.remove_node(v)


def _components(g):
    return [g.subgraph(c).copy() for c in nx.weakly_connected_components(g)]


def _phase_graph(ctg_G, p_asm_G, h_asm_G, arid_to_phase):
    """[53-201] primary edges, phased-assembly edges, hooks at dead ends -- each with its reverse-strand mirror"""
    sg = nx.DiGraph()
    for v, w in ctg_G.edges():
        if p_asm_G.sg_edges[(v, w)][-1] != "G":
            continue
        pv, pw = arid_to_phase.get(v[:9], (-1, 0)), arid_to_phase.get(w[:9], (-1, 0))
        cross = "Y" if (pv[0] == pw[0] and pv[1] != pw[1]) else "N"
        rw, rv = _mirror(v, w)
        for n, ph in ((v, pv), (w, pw)):
            sg.add_node(n, label=_tag(ph), phase=_tag(ph), src="P")
        sg.add_edge(v, w, src="OP", cross_phase=cross)
        for n, ph in ((rv, pv), (rw, pw)):
            sg.add_node(n, label=_tag(ph), phase=_tag(ph), src="P")
        sg.add_edge(rw, rv, src="OP", cross_phase=cross)
    primary_nodes, primary_edges = set(sg.nodes()), set(sg.edges())

    def eligible(v, w):
        if v[:9] not in arid_to_phase or w[:9] not in arid_to_phase:
            return False
        return not ((v, w) in primary_edges and p_asm_G.sg_edges[(v, w)][-1] == "G")

    def add_pair(v, w, src, cross):
        pv, pw = arid_to_phase[v[:9]], arid_to_phase[w[:9]]
        rw, rv = _mirror(v, w)
        for n, ph in ((v, pv), (w, pw)):
            if n not in primary_nodes:
                sg.add_node(n, label=_tag(ph), phase=_tag(ph), src="H")
        sg.add_edge(v, w, src=src, cross_phase=cross)
        for n, ph in ((rv, pv), (rw, pw)):
            if n not in primary_nodes:
                sg.add_node(n, label=_tag(ph), phase=_tag(ph), src="H")
        sg.add_edge(rw, rv, src=src, cross_phase=cross)
    for (v, w), data in h_asm_G.sg_edges.items():
        if eligible(v, w) and data[-1] == "G":
            add_pair(v, w, "H", "N")
    before = sg.copy()
    for (v, w) in h_asm_G.sg_edges:
        if not eligible(v, w):
            continue
        if before.in_degree(w) == 0:
            add_pair(v, w, "ext", "Y")
        if before.out_degree(v) == 0:
            add_pair(v, w, "ext", "Y")
    return sg


def _best_path(G, source, target, weight=None):
    """The reference's nx.shortest_path calls (graphs_to_h_tigs.py:238, 243, 354-356, 505) with the choice among equally cheap paths made
    explicit, so that it does not depend on the networkx version at hand (1.x: heap ties fall to node comparison / adjacency hash order of
    Python 2 -- unspecified; 2.x-3.x: first discovered, i.e. edge insertion order).  Pinned rule: least total weight (1 per edge without a
    weight key); among those, fewest edges; among those, walking back from the target, the predecessor with the smallest name at every step.
    Weights are the layout's integers (1 / 50 / 100000).  Raises nx.NetworkXNoPath like the call it replaces."""
    import heapq
    if source not in G or target not in G:
        raise nx.exception.NodeNotFound("node not in graph")
    best = {source: (0, 0)}
    heap = [(0, 0, source)]
    done = set()
    while heap:
        d, h, v = heapq.heappop(heap)
        if v in done:
            continue
        done.add(v)
        if v == target:
            break
        for _, w, data in G.out_edges(v, data=True):
            cand = (d + (data.get(weight, 1) if weight else 1), h + 1)
            if w not in best or cand < best[w]:
                best[w] = cand
                heapq.heappush(heap, (cand[0], cand[1], w))
    if target not in done:
        raise nx.exception.NetworkXNoPath("No path between %s and %s." % (source, target))
    path, v = [target], target
    while v != source:
        d, h = best[v]
        v = min(u for u, _, data in G.in_edges(v, data=True) if u in done and best[u] == (d - (data.get(weight, 1) if weight else 1), h - 1))
        path.append(v)
    return path[::-1]


def _prune_strand_crossers(sg, ctg_G):
    """[203-258] haplotype components must hang on the contig's own strand: keep only shortest hooks of components that touch
    both strands, drop components that touch neither"""
    ctg_nodes = set(ctg_G.nodes())
    ctg_nodes_r = set(reverse_end(v) for v in ctg_nodes)
    rest = sg.copy()
    for v, w in ctg_G.edges():
        rest.remove_edge(v, w)
        rest.remove_edge(*_mirror(v, w))
    _drop_isolated(rest)
    drop_nodes, drop_edges = set(), set()
    for comp in _components(rest):
        members = set(comp.nodes())
        fwd, rev = len(members & ctg_nodes) > 0, len(members & ctg_nodes_r) > 0
        if fwd and rev:
            on_ctg = lambda n: n in ctg_nodes or n in ctg_nodes_r
            sources = [n for n in comp.nodes() if comp.in_degree(n) == 0 or on_ctg(n)]
            sinks = [n for n in comp.nodes() if comp.out_degree(n) == 0 or on_ctg(n)]
            keep = set()
            for v in sources:
                for w in sinks:
                    path = []
                    if (v in ctg_nodes and w not in ctg_nodes_r) or (v not in ctg_nodes and w in ctg_nodes_r):
                        try:
                            path = _best_path(comp, v, w)
                        except nx.exception.NetworkXNoPath:
                            path = []
                    if len(path) >= 2:
                        for a, b in zip(path[:-1], path[1:]):
                            keep.add((a, b))
                            keep.add(_mirror(a, b))
            for v, w in comp.edges():
                if (v, w) not in keep:
                    drop_edges.add((v, w))
                    drop_edges.add(_mirror(v, w))
        if not fwd and not rev:
            drop_nodes.update(members)
            drop_nodes.update(reverse_end(v) for v in members)
    for v, w in list(drop_edges):
        sg.remove_edge(v, w)
    for v in drop_nodes:
        sg.remove_node(v)
    _drop_isolated(sg)


def _score_edges(sg):
    """[263-281]"""
    for v, w in sg.edges():
        p0, p1 = sg.nodes[v]["phase"].split("_"), sg.nodes[w]["phase"].split("_")
        if p0 == p1:
            sg[v][w].update(weight=10, score=1, label="type0")
        elif p0[0] == p1[0]:
            sg[v][w].update(weight=1, score=100000, label="type1")
        else:
            sg[v][w].update(weight=5, score=50, label="type2")


def _phase_consistent_view(sg):
    """[284-311] a copy without the hooks and without phase-switching edges that have an in-phase alternative at both ends"""
    g = sg.copy()
    same = lambda a, b: g.nodes[a]["phase"] == g.nodes[b]["phase"]
    drop = set()
    for v, w in g.edges():
        if g[v][w]["src"] == "ext":
            drop.add((v, w))
            drop.add(_mirror(v, w))
        if same(v, w):
            continue
        if any(same(a, b) for a, b in g.out_edges(v)) and any(same(a, b) for a, b in g.in_edges(w)):
            drop.add((v, w))
            drop.add(_mirror(v, w))
    for v, w in list(drop):
        g.remove_edge(v, w)
    return g


def _cut_shortcuts(sg, sg2, s_node, ctg_id, out_dir):
    """[314-347] an edge into a node that is reached much earlier than the node's other in-edges is a repeat-induced short cut
    (edges of the original contig are never cut)"""
    with open(os.path.join(out_dir, "path_len.%s" % ctg_id), "w") as f:
        dist = dict(nx.shortest_path_length(sg, source=s_node))
        cut = set()
        for w in dist:
            if sg.in_degree(w) < 2:
                continue
            longest = 0
            for v, _ in sg.in_edges(w):
                if v in dist and dist[v] > longest:
                    longest = dist[v]
            if longest == 0:
                continue
            for v, _ in sg.in_edges(w):
                if v in dist:
                    print(ctg_id, "link_lengths", v, w, longest, dist[v], file=f)
                    if longest - dist[v] > 10 and sg[v][w]["src"] != "OP":
                        cut.add((v, w))
                        cut.add(_mirror(v, w))
        for v, w in list(cut):
            sg.remove_edge(v, w)
            print(ctg_id, "removed", v, w, file=f)
        for v, w in list(cut):
            if sg2.has_edge(v, w):
                sg2.remove_edge(v, w)


def _emit_path(sg, path_edges, tig_name, p_asm_G, h_asm_G, arid_to_phase, seqs, f_edges, f_path):
    """the `*_edges` and `*_path` rows of one tig and its sequence pieces [365-397, 519-550]"""
    pieces = []
    for v, w in path_edges:
        sg[v][w]["h_edge"] = 1
        pv, pw = arid_to_phase.get(v.split(":")[0], (-1, 0)), arid_to_phase.get(w.split(":")[0], (-1, 0))
        print(tig_name, v, w, sg[v][w]["cross_phase"], sg[v][w]["src"], pv[0], pv[1], pw[0], pw[1], file=f_edges)
        data = p_asm_G.sg_edges[(v, w)] if sg[v][w]["src"] == "OP" else h_asm_G.sg_edges[(v, w)]
        seq_id, s, t = data[0]
        pieces.append(_edge_seq(seqs, data))
        print(tig_name, v, w, seq_id, s, t, data[1], data[2], "%d %d" % arid_to_phase.get(seq_id, (-1, 0)), file=f_path)
        sg[v][w]["tig_id"] = tig_name
    return "".join(pieces)


def _peel_haplotigs(rest):
    """[457-513] per component: repeatedly take the longest of the cheapest source-to-sink paths; -> {(s, t): path} in discovery order"""
    labelled, h_paths = set(), {}
    for comp in _components(rest):
        sub = comp.copy()
        while sub.size() > 5:
            sources = [n for n in sub.nodes() if sub.in_degree(n) != 1]
            sinks = [n for n in sub.nodes() if sub.out_degree(n) != 1]
            if not sources and not sinks:
                break                                    # only cycles are left
            longest, dead_sinks = [], set()
            for s in sources:
                if s in labelled:
                    continue
                found = []
                for t in sinks:
                    if t in dead_sinks:
                        continue
                    try:
                        found.append((_best_path(sub, s, t, weight="score"), t))
                    except nx.exception.NetworkXNoPath:
                        continue
                found.sort(key=lambda x: -len(x[0]))
                if not found:
                    continue
                if len(found[0][0]) > len(longest):
                    longest = found[0][0]
                for _, t in found[1:]:
                    dead_sinks.add(t)
            if not longest:
                break
            h_paths[(longest[0], longest[-1])] = longest
            labelled.add(longest[0])
            labelled.add(reverse_end(longest[0]))
            for v in longest:
                sub.remove_node(v)
    return h_paths


def generate_haplotigs_for_ctg(ctg_id, out_dir, p_asm_G, h_asm_G, arid_to_phase, seqs):
    """one contig: graphs_to_h_tigs.py:38-593"""
    os.makedirs(out_dir, exist_ok=True)
    ctg_G = p_asm_G.get_sg_for_ctg(ctg_id)
    sg = _phase_graph(ctg_G, p_asm_G, h_asm_G, arid_to_phase)
    _prune_strand_crossers(sg, ctg_G)
    s_node = p_asm_G.ctg_data[ctg_id][5][0][0]
    t_node = p_asm_G.ctg_data[ctg_id][5][-1][-1]
    _score_edges(sg)
    sg2 = _phase_consistent_view(sg)
    _cut_shortcuts(sg, sg2, s_node, ctg_id, out_dir)
    nx.write_gexf(sg, os.path.join(out_dir, "sg.gexf"))
    nx.write_gexf(sg2, os.path.join(out_dir, "sg2.gexf"))
    try:
        s_path = _best_path(sg2, s_node, t_node, weight="score")
    except nx.exception.NetworkXNoPath:
        s_path = _best_path(sg, s_node, t_node, weight="score")
    s_path_edges = list(zip(s_path[:-1], s_path[1:]))
    for v, w in s_path_edges:
        sg[v][w]["weight"] = 15
    # ---- the updated primary contig [361-405]
    with open(os.path.join(out_dir, "p_ctg_path.%s" % ctg_id), "w") as f_path, open(os.path.join(out_dir, "p_ctg.%s.fa" % ctg_id), "w") as f_fa, \
            open(os.path.join(out_dir, "p_ctg_edges.%s" % ctg_id), "w") as f_edges:
        seq = _emit_path(sg, s_path_edges, "%s" % ctg_id, p_asm_G, h_asm_G, arid_to_phase, seqs, f_edges, f_path)
        print(">%s" % ctg_id, file=f_fa)
        print(seq, file=f_fa)
    used = set(s_path_edges) | set(_mirror(v, w) for v, w in s_path_edges)
    # ---- what the primary path left over [409-455]
    rest = sg.copy()
    from_s = nx.descendants(rest, s_node)
    to_t = nx.descendants(rest.reverse(), t_node)
    for v, w in list(used):
        rest.remove_edge(v, w)
    for v, w in list(rest.edges()):
        if rest[v][w]["cross_phase"] == "Y":
            rest.remove_edge(v, w)
    for v in list(rest.nodes()):
        if v not in (from_s | to_t):
            rest.remove_node(v)
    for v in list(rest.nodes()):
        if rest.out_degree(v) == 0 and rest.in_degree(v) == 0:
            rest.remove_node(v)
            continue
        rest.nodes[v]["reachable"] = 1 if v in (from_s & to_t) else 0
    for v in p_asm_G.get_sg_for_ctg(ctg_id).nodes():
        rv = reverse_end(v)
        if rv in rest:
            rest.remove_node(rv)
    # ---- haplotigs [457-560]
    h_paths = _peel_haplotigs(rest)
    with open(os.path.join(out_dir, "h_ctg_path.%s" % ctg_id), "w") as f_path, open(os.path.join(out_dir, "h_ctg_all.%s.fa" % ctg_id), "w") as f_fa, \
            open(os.path.join(out_dir, "h_ctg_edges.%s" % ctg_id), "w") as f_edges:
        for h_tig_id, path in enumerate(h_paths.values(), start=1):
            name = "%s_%03d" % (ctg_id, h_tig_id)
            seq = _emit_path(sg, list(zip(path[:-1], path[1:])), name, p_asm_G, h_asm_G, arid_to_phase, seqs, f_edges, f_path)
            print(">%s" % name, file=f_fa)
            print(seq, file=f_fa)
    return len(h_paths)


def load_rid_to_phase(path):
    """`rid_to_phase.all` rows 'pread ctg block phase' (phasing_readmap.py:47-51, unzip.py:303-314) -> {ctg: {pread: (block, phase)}}, all ids"""
    table, ids = {}, set()
    with open(path) as f:
        for row in f:
            row = row.strip().split()
            table.setdefault(row[1], {})[row[0]] = (int(row[2]), int(row[3]))
            ids.add(row[0])
    return table, ids


def parse_args(argv):
    parser = argparse.ArgumentParser(description='layout haplotigs from primary assembly graph and phased aseembly graph')
    parser.add_argument('--fc_asm_path', type=str, help='path to the primary Falcon assembly output directory', required=True)
    parser.add_argument('--fc_hasm_path', type=str, help='path to the phased Falcon assembly output directory', required=True)
    parser.add_argument('--ctg_id', type=str, help='contig identifier in the bam file', default="all", required=True)
    parser.add_argument('--base_dir', type=str, default="./", help='the output base_dir, default to current working directory')
    parser.add_argument('--rid_phase_map', type=str, help="path to the file that encode the relationship of the read id to phase blocks", required=True)
    parser.add_argument('--fasta', type=str, help="sequence file of the p-reads", required=True)
    return parser.parse_args(argv[1:])


def main(argv=sys.argv):
    args = parse_args(argv)
    p_asm_G = AsmGraph(os.path.join(args.fc_asm_path, "sg_edges_list"), os.path.join(args.fc_asm_path, "utg_data"), os.path.join(args.fc_asm_path, "ctg_paths"))
    h_asm_G = AsmGraph(os.path.join(args.fc_hasm_path, "sg_edges_list"), os.path.join(args.fc_hasm_path, "utg_data"), os.path.join(args.fc_hasm_path, "ctg_paths"))
    all_rid_to_phase, read_ids = load_rid_to_phase(args.rid_phase_map)
    for g in (p_asm_G, h_asm_G):                      # reads on kept ('G') edges of either graph [646-660]
        for (v, w), data in g.sg_edges.items():
            if data[-1] == "G":
                read_ids.add(v.split(":")[0])
                read_ids.add(w.split(":")[0])
    seqs = load_sg_seq(read_ids, args.fasta)
    ctg_ids = list(p_asm_G.ctg_data.keys()) if args.ctg_id == "all" else [args.ctg_id]
    done = []
    for ctg_id in ctg_ids:
        if ctg_id[-1] != "F" or ctg_id not in all_rid_to_phase:
            continue
        # the reference writes to ./<ctg_id> whatever --base_dir says (its :672); kept
        generate_haplotigs_for_ctg(ctg_id, os.path.join(".", ctg_id), p_asm_G, h_asm_G, all_rid_to_phase[ctg_id], seqs)
        done.append(ctg_id)
    return done
