#!/usr/bin/env python3
"""Generate the 256-case marching-cubes triangle table from first principles.

No table is copied from anywhere: for every corner-sign configuration the
iso-surface polygons are found by tracing directed segments over the six cube
faces (inside region kept on the left when the face is seen from outside the
cube), chaining them into closed loops and fan-triangulating each loop.  The
ambiguous face (two diagonal inside corners) is resolved by a RULE that depends
only on the four face-corner signs, so two cubes sharing a face always agree
and the mesh is crack free:
  rule 0 (mc_table.inc)    each inside corner is cut off separately
  rule 1 (mc_table_r1.inc) each outside corner is cut off separately (inside corners joined)
A third table reproduces the behaviour of the classic published table, which is
complement-symmetric instead of face-consistent (and therefore can crack):
  rule 2 (mc_table_r2.inc) rule 0 for cases with <= 4 inside corners, else rule 1
(nvbx_mapper_params::mesh_ambiguity_rule; [U] which of these the reference's
table implements cannot be read off /root/reference -- the core is absent.)

Corner i is "inside" when bit i of the case index is set (distance < 0).
Triangles are wound so that their normal points to the positive-distance
(outside / free-space) side.

Conventions (shared by oracle/ and csrc/):
  corners: 0:(0,0,0) 1:(1,0,0) 2:(1,1,0) 3:(0,1,0) 4:(0,0,1) 5:(1,0,1) 6:(1,1,1) 7:(0,1,1)
  edges:   0:0-1 1:1-2 2:2-3 3:3-0 4:4-5 5:5-6 6:6-7 7:7-4 8:0-4 9:1-5 10:2-6 11:3-7

Usage: gen_mc_table.py OUT.inc [OUT2.inc ...]
"""
import sys
import numpy as np

CORNERS = np.array([(0,0,0),(1,0,0),(1,1,0),(0,1,0),(0,0,1),(1,0,1),(1,1,1),(0,1,1)], float)
EDGES = [(0,1),(1,2),(2,3),(3,0),(4,5),(5,6),(6,7),(7,4),(0,4),(1,5),(2,6),(3,7)]
EDGE_ID = {}
for i,(a,b) in enumerate(EDGES):
    EDGE_ID[(a,b)] = i; EDGE_ID[(b,a)] = i
# faces: corner cycles (any order; geometry decides orientation) + outward normal
FACES = [((0,1,2,3),(0,0,-1)), ((4,5,6,7),(0,0,1)), ((0,1,5,4),(0,-1,0)),
         ((3,2,6,7),(0,1,0)), ((0,3,7,4),(-1,0,0)), ((1,2,6,5),(1,0,0))]

def mid(e):
    a,b = EDGES[e]
    return 0.5*(CORNERS[a]+CORNERS[b])

def directed(ea, eb, p_in, n):
    """Return (from,to) so that p_in lies to the left of the segment seen from outside."""
    d = mid(eb)-mid(ea)
    left = np.dot(np.cross(np.array(n,float), d), p_in-mid(ea))
    return (ea,eb) if left > 0 else (eb,ea)

def case_loops(case, rule=0):
    inside = [(case>>i)&1 for i in range(8)]
    nxt = {}
    for cyc,n in FACES:
        ins = [c for c in cyc if inside[c]]
        if len(ins) in (0,4):
            continue
        cross = []
        for k in range(4):
            a,b = cyc[k], cyc[(k+1)%4]
            if inside[a] != inside[b]:
                cross.append(EDGE_ID[(a,b)])
        segs = []
        if len(cross) == 2:
            segs.append((cross[0],cross[1],CORNERS[ins[0]]))
        else:  # 4 crossings: diagonal inside corners, cut each off separately
            assert len(cross)==4 and len(ins)==2
            if rule == 0:
                for c in ins:
                    k = cyc.index(c)
                    e1 = EDGE_ID[(cyc[(k-1)%4],c)]; e2 = EDGE_ID[(c,cyc[(k+1)%4])]
                    segs.append((e1,e2,CORNERS[c]))
            else:      # cut off each OUTSIDE corner: the inside region (it contains the face centre) joins the two inside corners
                centre = sum(CORNERS[c] for c in cyc) / 4.0
                for c in cyc:
                    if inside[c]: continue
                    k = cyc.index(c)
                    e1 = EDGE_ID[(cyc[(k-1)%4],c)]; e2 = EDGE_ID[(c,cyc[(k+1)%4])]
                    segs.append((e1,e2,centre))
        for ea,eb,p in segs:
            f,t = directed(ea,eb,p,n)
            assert f not in nxt, "edge has two outgoing segments"
            nxt[f] = t
    loops = []
    seen = set()
    for start in sorted(nxt):
        if start in seen: continue
        loop = [start]; seen.add(start)
        cur = nxt[start]
        while cur != start:
            assert cur not in seen
            loop.append(cur); seen.add(cur); cur = nxt[cur]
        loops.append(loop)
    assert len(seen) == len(nxt)
    return loops

def build(rule=0):
    # decide global winding with case 1 (corner 0 inside): normal must point away from corner 0
    table = []
    flip = None
    for case in range(256):
        tris = []
        for loop in case_loops(case, rule):
            assert len(loop) >= 3
            for k in range(1,len(loop)-1):
                tris.append((loop[0],loop[k],loop[k+1]))
        table.append(tris)
    a,b,c = table[1][0]
    nrm = np.cross(mid(b)-mid(a), mid(c)-mid(a))
    flip = np.dot(nrm, np.array([1.,1.,1.])) < 0
    if flip:
        table = [[(a,c,b) for (a,b,c) in tris] for tris in table]
    # sanity: every case, every triangle normal has positive dot with (outside centroid - inside centroid)
    for case in range(1,255):
        ins = [i for i in range(8) if (case>>i)&1]; out=[i for i in range(8) if not (case>>i)&1]
        assert len(table[case]) <= 5, (case, len(table[case]))
    return table

def emit(table, title):
    lines = []
    lines.append("/* GENERATED by tools/gen_mc_table.py -- do not edit. 256 x 16 edge ids, -1 terminated. %s */" % title)
    lines.append("/* corners: 0:(0,0,0) 1:(1,0,0) 2:(1,1,0) 3:(0,1,0) 4:(0,0,1) 5:(1,0,1) 6:(1,1,1) 7:(0,1,1) */")
    lines.append("/* edges: 0:0-1 1:1-2 2:2-3 3:3-0 4:4-5 5:5-6 6:6-7 7:7-4 8:0-4 9:1-5 10:2-6 11:3-7 */")
    for case,tris in enumerate(table):
        flat = [e for t in tris for e in t]
        flat += [-1]*(16-len(flat))
        lines.append("{" + ",".join("%2d"%v for v in flat) + "},")
    return "\n".join(lines)+"\n"

def main():
    import os
    t0 = build(0); t1 = build(1)
    t2 = [t0[c] if bin(c).count("1") <= 4 else t1[c] for c in range(256)]
    texts = {"mc_table.inc": emit(t0, "rule 0"), "mc_table_r1.inc": emit(t1, "rule 1"), "mc_table_r2.inc": emit(t2, "rule 2")}
    # arguments: the rule-0 file of every tree (oracle/mc_table.inc, csrc/mc_table.inc); the other rules go beside it
    for p in sys.argv[1:]:
        d = os.path.dirname(p)
        for name, txt in texts.items():
            q = os.path.join(d, name)
            if os.path.exists(q) and open(q).read() == txt:
                continue          # keep mtimes stable so make does not rebuild
            open(q,"w").write(txt)
    print("max tris", max(len(t) for t in t0 + t1 + t2), "total", sum(len(t) for t in t0), sum(len(t) for t in t1), sum(len(t) for t in t2))

if __name__ == "__main__":
    main()
