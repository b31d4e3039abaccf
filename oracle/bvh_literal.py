"""Literal transcription of the reference's BVH builder -- TEST INFRASTRUCTURE.

Follows src/passes/raytrace.ts:540-694 statement by statement, including the O(n^2)
re-scan of every split candidate (:626-643) and the indexOf-based flatten (:687-688),
in pure-Python floats (IEEE doubles, like JavaScript numbers).  Only usable for a few
hundred triangles; tests compare the product's fast builder (mi3pt_host_build_bvh_f64)
against it for tree identity.  PARITY UNPINNED: the reference has no tests for this.
"""
import math


class Box3:
    """THREE.Box3 subset (min/max as lists)."""

    def __init__(self):
        self.min = [math.inf, math.inf, math.inf]
        self.max = [-math.inf, -math.inf, -math.inf]

    def expand_by_point(self, p):
        for k in range(3):
            self.min[k] = min(self.min[k], p[k])
            self.max[k] = max(self.max[k], p[k])
        return self

    def set_from_points(self, pts):
        for p in pts:
            self.expand_by_point(p)
        return self

    def is_empty(self):
        return self.max[0] < self.min[0] or self.max[1] < self.min[1] or self.max[2] < self.min[2]

    def get_size(self):
        if self.is_empty():
            return [0.0, 0.0, 0.0]
        return [self.max[k] - self.min[k] for k in range(3)]

    def get_center(self):
        if self.is_empty():
            return [0.0, 0.0, 0.0]
        return [(self.min[k] + self.max[k]) * 0.5 for k in range(3)]


class Node:
    def __init__(self, bbox, is_leaf, triangle_index=-1):
        self.bbox = bbox
        self.is_leaf = is_leaf
        self.triangle_index = triangle_index
        self.left = None
        self.right = None


def build_bvh(positions):
    """raytrace.ts:540-560.  positions: sequence of triangles, each ((ax,ay,az),(b..),(c..))."""
    input_nodes = []
    for i, t in enumerate(positions):
        bbox = Box3().set_from_points([list(map(float, t[0])), list(map(float, t[1])), list(map(float, t[2]))])
        input_nodes.append(Node(bbox, True, i))
    return build_bvh_recursive(input_nodes)


def _compute_bbox(nodes):
    bbox = Box3()
    for node in nodes:
        bbox.expand_by_point(node.bbox.min)
        bbox.expand_by_point(node.bbox.max)
    return bbox


def _surface_area(box):
    x, y, z = box.get_size()
    return 2 * (x * y + x * z + y * z)


def build_bvh_recursive(input_nodes):
    """raytrace.ts:562-655"""
    if len(input_nodes) == 0:
        raise ValueError("Input nodes array is empty")
    if len(input_nodes) == 1:
        return input_nodes[0]
    node = Node(Box3(), False)
    for inp in input_nodes:
        node.bbox.expand_by_point(inp.bbox.min)
        node.bbox.expand_by_point(inp.bbox.max)
    if len(input_nodes) == 2:
        node.left, node.right = input_nodes[0], input_nodes[1]
        return node
    size = node.bbox.get_size()
    axis = (0 if size[0] > size[2] else 2) if size[0] > size[1] else 1
    # Array.prototype.sort is stable; the comparator is centerA - centerB
    input_nodes.sort(key=lambda n: n.bbox.get_center()[axis])
    min_cost = math.inf
    min_index = -1
    for i in range(1, len(input_nodes)):
        left_nodes = input_nodes[:i]
        right_nodes = input_nodes[i:]
        left_area = _surface_area(_compute_bbox(left_nodes))
        right_area = _surface_area(_compute_bbox(right_nodes))
        cost = left_area * len(left_nodes) + right_area * len(right_nodes)
        if cost < min_cost:
            min_cost = cost
            min_index = i
    left_nodes = input_nodes[:min_index]
    right_nodes = input_nodes[min_index:]
    node.left = build_bvh_recursive(left_nodes)
    node.right = build_bvh_recursive(right_nodes)
    return node


def flatten_bvh(root):
    """raytrace.ts:667-694; returns a list of dicts (min, max, isLeaf, left, right, triangleIndex)."""
    nodes = []
    queue = [root]
    while queue:
        node = queue.pop(0)
        nodes.append(node)
        if not node.is_leaf:
            queue.append(node.left)
            queue.append(node.right)
    index_of = {id(n): i for i, n in enumerate(nodes)}     # nodes.indexOf(node.left)
    flat = []
    for node in nodes:
        flat.append(dict(min=list(node.bbox.min), max=list(node.bbox.max),
                         isLeaf=1 if node.is_leaf else 0,
                         left=-1 if node.is_leaf else index_of[id(node.left)],
                         right=-1 if node.is_leaf else index_of[id(node.right)],
                         triangleIndex=node.triangle_index if node.is_leaf else -1))
    return flat
