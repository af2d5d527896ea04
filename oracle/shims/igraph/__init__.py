"""
ORACLE / TEST INFRASTRUCTURE ONLY -- never imported by the product path.

Minimal pure-Python stand-in for ``python-igraph==0.8.2`` (requirements.txt:4 of the reference), which is a
third-party dependency that is NOT vendored under /root/reference and NOT installed in this image. It restates
only the API surface the reference touches, all of it inside graph_ltpl/data_objects/GraphBase.py (see
SURVEY.md appendix C for the call-site list). Parity at this boundary is UNPINNED by the reference (it ships
no tests / golden vectors); the Dijkstra below is cross-checked against scipy/networkx in tests/.

Semantics restated (GraphBase.py line numbers):
  * directed multigraph with per-vertex / per-edge attribute dicts        (122-123, 163-194, 409-416)
  * vertex lookup by ``name``; ``ValueError`` when absent                 (255, 883, 917)
  * ``get_eid(a, b, error=False)`` -> -1 when missing, else ValueError    (407, 417, 468, 508, 538, 564)
  * ``get_eids(pairs=[[a, b], ...])`` duplicates allowed                  (772)
  * ``delete_edges(eid | [eids])`` duplicates tolerated                   (565, 775)
  * ``vs.select(layer_id_ge/le/in=..., name_notin=...)``                  (615, 619, 705, 709, 744)
  * ``induced_subgraph(vertexseq)``: vertices keep relative order,
    attributes copied, only edges with both ends survive                  (621, 711, 745)
  * ``copy()`` deep enough that ``es[e]['offline_cost'] *= f`` on the
    copy does not leak into the source graph                              (478-512, 672, 774)
  * ``successors`` / ``predecessors``                                     (266, 274)
  * ``get_shortest_paths(v, to=, weights=, output="vpath")``: Dijkstra,
    strict '<' relaxation (igraph 0.8 semantics); ties between equally
    distant heap entries are popped in vertex-index order -- this is the
    tie rule the whole project pins (SURVEY.md §8a tie-break note).        (818-821)
"""

import heapq

__version__ = "0.8.2-oracle-shim"


class _Vertex(object):
    __slots__ = ("_g", "index")

    def __init__(self, g, index):
        self._g = g
        self.index = index

    def __getitem__(self, key):
        return self._g._vattr[key][self.index] if key in self._g._vattr else None

    def __setitem__(self, key, value):
        self._g._vcol(key)[self.index] = value


class _Edge(object):
    __slots__ = ("_g", "index")

    def __init__(self, g, index):
        self._g = g
        self.index = index

    @property
    def source(self):
        return self._g._esrc[self.index]

    @property
    def target(self):
        return self._g._etgt[self.index]

    @property
    def tuple(self):
        return self._g._esrc[self.index], self._g._etgt[self.index]

    def __getitem__(self, key):
        return self._g._eattr[key][self.index] if key in self._g._eattr else None

    def __setitem__(self, key, value):
        self._g._ecol(key)[self.index] = value


class _VertexSeq(object):
    def __init__(self, g, indices=None):
        self._g = g
        self._idx = indices  # None -> all vertices

    def _indices(self):
        return range(self._g._nv) if self._idx is None else self._idx

    def __len__(self):
        return self._g._nv if self._idx is None else len(self._idx)

    def __iter__(self):
        g = self._g
        for i in self._indices():
            yield _Vertex(g, i)

    def __getitem__(self, key):
        if isinstance(key, str):
            col = self._g._vattr.get(key)
            if col is None:
                return [None for _ in self._indices()]
            return [col[i] for i in self._indices()]
        key = int(key)
        if self._idx is not None:
            return _Vertex(self._g, self._idx[key])
        if key < 0:
            key += self._g._nv
        if not 0 <= key < self._g._nv:
            raise IndexError("vertex index out of range")
        return _Vertex(self._g, key)

    def find(self, name):
        idx = self._g._name_index().get(name)
        if idx is None:
            raise ValueError("no such vertex: %r" % (name,))
        return _Vertex(self._g, idx)

    def select(self, **kwargs):
        g = self._g
        cand = list(self._indices())
        for key, val in kwargs.items():
            attr, _, op = key.rpartition("_")
            col = g._vattr.get(attr)
            if col is None:
                col = [None] * g._nv
            if op == "ge":
                cand = [i for i in cand if col[i] >= val]
            elif op == "le":
                cand = [i for i in cand if col[i] <= val]
            elif op == "in":
                vs = set(val)
                cand = [i for i in cand if col[i] in vs]
            elif op == "notin":
                vs = set(val)
                cand = [i for i in cand if col[i] not in vs]
            elif op == "eq":
                cand = [i for i in cand if col[i] == val]
            else:
                raise NotImplementedError("vertex select operator '%s' not restated in the oracle shim" % key)
        return _VertexSeq(g, cand)


class _EdgeSeq(object):
    def __init__(self, g, indices=None):
        self._g = g
        self._idx = indices

    def _indices(self):
        if self._idx is not None:
            return self._idx
        return self._g._live_edges()

    def __len__(self):
        return len(self._indices())

    def __iter__(self):
        g = self._g
        for e in self._indices():
            yield _Edge(g, e)

    def __call__(self, key):
        # GraphBase.py:469 -- ``self.__g.es(edge_id)`` selects a one-element sequence
        if isinstance(key, (list, tuple)):
            return _EdgeSeq(self._g, [int(k) for k in key])
        return _EdgeSeq(self._g, [int(key)])

    def __getitem__(self, key):
        if isinstance(key, str):
            col = self._g._eattr.get(key)
            if col is None:
                return [None for _ in self._indices()]
            return [col[e] for e in self._indices()]
        key = int(key)
        if self._idx is not None:
            return _Edge(self._g, self._idx[key])
        self._g._check_eid(key)
        return _Edge(self._g, key)


class Graph(object):
    """Directed multigraph with attribute columns (only the surface used by GraphBase.py)."""

    def __init__(self, directed=False):
        self._directed = directed
        self._nv = 0
        self._vattr = {}        # attr -> list (len nv)
        self._esrc = []         # edge -> source vertex index
        self._etgt = []
        self._ealive = []       # tombstones (ids stay stable until a copy / subgraph compacts them)
        self._eattr = {}        # attr -> list (len ne incl. tombstones)
        self._out = []          # vertex -> list of edge ids
        self._in = []
        self._names = None      # lazy name -> vertex index
        self._pairs = None      # lazy (src, tgt) -> first live edge id
        self._n_dead = 0

    # ---- helpers ----------------------------------------------------------------------------------------------
    def _vcol(self, key):
        col = self._vattr.get(key)
        if col is None:
            col = [None] * self._nv
            self._vattr[key] = col
        return col

    def _ecol(self, key):
        col = self._eattr.get(key)
        if col is None:
            col = [None] * len(self._esrc)
            self._eattr[key] = col
        return col

    def _name_index(self):
        if self._names is None:
            names = self._vattr.get("name", [])
            self._names = {}
            for i, n in enumerate(names):
                if n is not None and n not in self._names:
                    self._names[n] = i
        return self._names

    def _pair_index(self):
        if self._pairs is None:
            self._pairs = {}
            for e in range(len(self._esrc)):
                if self._ealive[e]:
                    self._pairs.setdefault((self._esrc[e], self._etgt[e]), e)
        return self._pairs

    def _live_edges(self):
        if self._n_dead == 0:
            return range(len(self._esrc))
        return [e for e in range(len(self._esrc)) if self._ealive[e]]

    def _check_eid(self, e):
        if not (0 <= e < len(self._esrc)) or not self._ealive[e]:
            raise ValueError("no such edge id: %r" % (e,))

    def _vid(self, v):
        if isinstance(v, _Vertex):
            return v.index
        if isinstance(v, str):
            idx = self._name_index().get(v)
            if idx is None:
                raise ValueError("no such vertex: %r" % (v,))
            return idx
        return int(v)

    # ---- construction -----------------------------------------------------------------------------------------
    def to_directed(self, *args, **kwargs):
        self._directed = True

    def is_directed(self):
        return self._directed

    def vcount(self):
        return self._nv

    def ecount(self):
        return len(self._esrc) - self._n_dead

    def add_vertex(self, name=None, **kwds):
        idx = self._nv
        self._nv += 1
        for col in self._vattr.values():
            col.append(None)
        self._out.append([])
        self._in.append([])
        if name is not None:
            kwds["name"] = name
        for k, v in kwds.items():
            self._vcol(k)[idx] = v
        if self._names is not None and name is not None:
            self._names.setdefault(name, idx)

    def add_edge(self, source, target, **kwds):
        s = self._vid(source)
        t = self._vid(target)
        e = len(self._esrc)
        self._esrc.append(s)
        self._etgt.append(t)
        self._ealive.append(True)
        for col in self._eattr.values():
            col.append(None)
        for k, v in kwds.items():
            self._ecol(k)[e] = v
        self._out[s].append(e)
        self._in[t].append(e)
        if self._pairs is not None:
            self._pairs.setdefault((s, t), e)

    # ---- sequences --------------------------------------------------------------------------------------------
    @property
    def vs(self):
        return _VertexSeq(self)

    @property
    def es(self):
        return _EdgeSeq(self)

    # ---- edge lookup / deletion -------------------------------------------------------------------------------
    def get_eid(self, v1, v2, directed=True, error=True):
        try:
            s = self._vid(v1)
            t = self._vid(v2)
        except ValueError:
            if error:
                raise
            return -1
        e = self._pair_index().get((s, t))
        if e is None:
            if error:
                raise ValueError("no such edge: %r -> %r" % (v1, v2))
            return -1
        return e

    def get_eids(self, pairs=None, directed=True, error=True):
        return [self.get_eid(a, b, error=error) for a, b in (pairs or [])]

    def delete_edges(self, edges):
        if isinstance(edges, (int,)) or not hasattr(edges, "__iter__"):
            edges = [edges]
        for e in set(int(x) for x in edges):
            self._check_eid(e)
            self._ealive[e] = False
            self._n_dead += 1
            s, t = self._esrc[e], self._etgt[e]
            self._out[s].remove(e)
            self._in[t].remove(e)
        self._pairs = None

    # ---- neighbourhood ----------------------------------------------------------------------------------------
    def successors(self, vertex):
        return [self._etgt[e] for e in self._out[self._vid(vertex)]]

    def predecessors(self, vertex):
        return [self._esrc[e] for e in self._in[self._vid(vertex)]]

    # ---- copies -----------------------------------------------------------------------------------------------
    def _subgraph(self, keep):
        """keep: sorted list of vertex indices (or None for all). Compacts tombstoned edges."""
        g = Graph(directed=self._directed)
        if keep is None:
            g._nv = self._nv
            g._vattr = {k: list(col) for k, col in self._vattr.items()}
            edges = self._live_edges()
            g._esrc = [self._esrc[e] for e in edges]
            g._etgt = [self._etgt[e] for e in edges]
        else:
            remap = {}
            for new, old in enumerate(keep):
                remap[old] = new
            g._nv = len(keep)
            g._vattr = {k: [col[i] for i in keep] for k, col in self._vattr.items()}
            edges = []
            esrc = self._esrc
            etgt = self._etgt
            g_esrc = g._esrc
            g_etgt = g._etgt
            for old in keep:
                for e in self._out[old]:
                    t = remap.get(etgt[e])
                    if t is not None:
                        edges.append(e)
            edges.sort()   # keep original relative edge order
            for e in edges:
                g_esrc.append(remap[esrc[e]])
                g_etgt.append(remap[etgt[e]])
        g._ealive = [True] * len(g._esrc)
        g._eattr = {k: [col[e] for e in edges] for k, col in self._eattr.items()}
        g._out = [[] for _ in range(g._nv)]
        g._in = [[] for _ in range(g._nv)]
        for e in range(len(g._esrc)):
            g._out[g._esrc[e]].append(e)
            g._in[g._etgt[e]].append(e)
        return g

    def copy(self):
        return self._subgraph(None)

    def induced_subgraph(self, vertices, implementation="auto"):
        if isinstance(vertices, _VertexSeq):
            keep = sorted(set(vertices._indices()))
        else:
            keep = sorted(set(self._vid(v) for v in vertices))
        return self._subgraph(keep)

    subgraph = induced_subgraph

    # ---- shortest path ----------------------------------------------------------------------------------------
    def get_shortest_paths(self, v, to=None, weights=None, mode=None, output="vpath"):
        """
        Dijkstra, binary heap keyed by (distance, vertex index); a vertex's parent changes only on a STRICTLY smaller
        tentative distance (igraph 0.8 ``igraph_get_shortest_paths_dijkstra`` behaviour). Out-edges of a vertex are
        relaxed in (target, edge id) order like igraph's indexed edge list.
        """
        if output != "vpath":
            raise NotImplementedError("only output='vpath' is restated in the oracle shim")
        src = self._vid(v)
        if to is None:
            targets = list(range(self._nv))
        elif isinstance(to, (list, tuple)):
            targets = [self._vid(t) for t in to]
        else:
            targets = [self._vid(to)]
        if weights is None:
            w = None
        elif isinstance(weights, str):
            w = self._eattr[weights]
        else:
            w = list(weights)

        inf = float("inf")
        dist = [inf] * self._nv
        parent = [-1] * self._nv
        done = [False] * self._nv
        dist[src] = 0.0
        heap = [(0.0, src)]
        remaining = set(targets)
        while heap and remaining:
            d, u = heapq.heappop(heap)
            if done[u]:
                continue
            done[u] = True
            remaining.discard(u)
            out = self._out[u]
            if len(out) > 1:
                out = sorted(out, key=lambda e: (self._etgt[e], e))
            for e in out:
                t = self._etgt[e]
                if done[t]:
                    continue
                alt = d + (1.0 if w is None else w[e])
                if alt < dist[t]:
                    dist[t] = alt
                    parent[t] = u
                    heapq.heappush(heap, (alt, t))

        res = []
        for t in targets:
            if t != src and parent[t] == -1:
                res.append([])
                continue
            path = [t]
            while path[-1] != src:
                path.append(parent[path[-1]])
            path.reverse()
            res.append(path)
        return res
