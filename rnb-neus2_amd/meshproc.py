"""Minimal triangle-mesh utilities for the pipeline's last stage and for albedo scaling: OBJ read/write (the layout
save_mesh emits, src/marching_cubes.cu:922-981), connected components, consistent outward orientation — the trimesh
calls of rnb_neus2/pipeline.py:178-219 and albedo_scaling.py:262-263 without trimesh."""
import numpy as np
from scipy import sparse
from scipy.sparse import csgraph


class Mesh:
    def __init__(self, vertices, faces, colors=None):
        self.vertices = np.asarray(vertices, np.float64).reshape(-1, 3)
        self.faces = np.asarray(faces, np.int64).reshape(-1, 3)
        self.colors = None if colors is None else np.asarray(colors, np.float64).reshape(-1, 3)

    @property
    def area(self):
        v = self.vertices[self.faces]
        return 0.5 * np.linalg.norm(np.cross(v[:, 1] - v[:, 0], v[:, 2] - v[:, 0]), axis=1).sum()

    @property
    def signed_volume(self):
        v = self.vertices[self.faces]
        return np.einsum("ij,ij->i", v[:, 0], np.cross(v[:, 1], v[:, 2])).sum() / 6.0

    def vertex_normals(self):
        v = self.vertices[self.faces]
        fn = np.cross(v[:, 1] - v[:, 0], v[:, 2] - v[:, 0])
        n = np.zeros_like(self.vertices)
        for k in range(3):
            np.add.at(n, self.faces[:, k], fn)
        l = np.linalg.norm(n, axis=1, keepdims=True)
        return n / np.where(l > 0, l, 1.0)

    def split(self):
        """Connected components (faces sharing a vertex), each with its own compact vertex array."""
        nf = len(self.faces)
        if nf == 0:
            return []
        nv = len(self.vertices)
        g = sparse.coo_matrix((np.ones(3 * nf), (np.repeat(np.arange(nf), 3), self.faces.ravel())), shape=(nf, nv)).tocsr()
        n_comp, labels = csgraph.connected_components(sparse.bmat([[None, g], [g.T, None]]), directed=False)
        labels = labels[:nf]
        out = []
        for c in np.unique(labels):
            f = self.faces[labels == c]
            used, inv = np.unique(f, return_inverse=True)
            out.append(Mesh(self.vertices[used], inv.reshape(-1, 3), None if self.colors is None else self.colors[used]))
        return out

    def fix_normals(self):
        """Make the winding consistent across shared edges (breadth-first per component), then flip components whose
        signed volume is negative so that normals point outward."""
        f = self.faces
        nf = len(f)
        if nf == 0:
            return
        e = np.concatenate([f[:, [0, 1]], f[:, [1, 2]], f[:, [2, 0]]])  # directed edges; edge k of face i is row i + k*nf
        owner = np.tile(np.arange(nf), 3)
        key = np.sort(e, axis=1)
        order = np.lexsort((key[:, 1], key[:, 0]))
        ks = key[order]
        same = np.all(ks[1:] == ks[:-1], axis=1)
        a, b = order[:-1][same], order[1:][same]  # pairs of half-edges on the same undirected edge
        flip_needed = e[a, 0] == e[b, 0]  # same direction in both faces => inconsistent winding
        fa, fb = owner[a], owner[b]
        adj = sparse.coo_matrix((np.ones(len(fa)), (fa, fb)), shape=(nf, nf)).tocsr()
        adj = adj + adj.T
        n_comp, labels = csgraph.connected_components(adj, directed=False)
        # parity propagation: x_b = x_a xor flip_needed along a spanning forest
        parity = {}
        for x, y, w in zip(fa, fb, flip_needed):
            parity[(x, y)] = w
            parity[(y, x)] = w
        flip = np.zeros(nf, bool)
        seen = np.zeros(nf, bool)
        indptr, indices = adj.indptr, adj.indices
        for seed in range(nf):
            if seen[seed]:
                continue
            seen[seed] = True
            queue = [seed]
            while queue:
                cur = queue.pop()
                for nb in indices[indptr[cur]:indptr[cur + 1]]:
                    if not seen[nb]:
                        seen[nb] = True
                        flip[nb] = flip[cur] ^ bool(parity[(cur, nb)])
                        queue.append(nb)
        self.faces = np.where(flip[:, None], f[:, ::-1], f)
        for c in range(n_comp):
            sel = labels == c
            v = self.vertices[self.faces[sel]]
            if np.einsum("ij,ij->i", v[:, 0], np.cross(v[:, 1], v[:, 2])).sum() < 0:
                self.faces[sel] = self.faces[sel][:, ::-1]


def load_obj(path):
    """`v x y z [r g b]` and `f a[/b[/c]] ...` records (polygons are fan-triangulated); other records are ignored."""
    verts, cols, faces = [], [], []
    with open(path) as f:
        for line in f:
            if line.startswith("v "):
                t = line.split()
                verts.append([float(t[1]), float(t[2]), float(t[3])])
                if len(t) >= 7:
                    cols.append([float(t[4]), float(t[5]), float(t[6])])
            elif line.startswith("f "):
                idx = [int(t.split("/")[0]) for t in line.split()[1:]]
                idx = [i - 1 if i > 0 else len(verts) + i for i in idx]
                for k in range(1, len(idx) - 1):
                    faces.append([idx[0], idx[k], idx[k + 1]])
    return Mesh(verts, faces, cols if len(cols) == len(verts) and cols else None)


def save_obj(path, mesh):
    n = mesh.vertex_normals()
    with open(path, "w") as f:
        if mesh.colors is not None:
            for v, c in zip(mesh.vertices, mesh.colors):
                f.write("v %.8f %.8f %.8f %.6f %.6f %.6f\n" % (v[0], v[1], v[2], c[0], c[1], c[2]))
        else:
            for v in mesh.vertices:
                f.write("v %.8f %.8f %.8f\n" % (v[0], v[1], v[2]))
        for q in n:
            f.write("vn %.8f %.8f %.8f\n" % (q[0], q[1], q[2]))
        for a, b, c in mesh.faces + 1:
            f.write("f %d//%d %d//%d %d//%d\n" % (a, a, b, b, c, c))
