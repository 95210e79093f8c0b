"""The bookkeeping of PBRS_FEAT_EXTENT (pbrs_amd/csrc/device/traverse.h, ClosestWalk::EXT) against the recursion it replaces, on random
trees.  tlas/src/bvh.rs:77-103 recurses: box test with ray.t_max, left subtree, `ray.set_extent(left result)` if there is one, right
subtree, the nearer of the two (`l.ray_t < r.ray_t ? l : r`).  The kernel walks with a stack: a pending right sibling carries, in a second
word, the smallest t returned since the window below it began; `win` is that of the innermost window; popping the sibling sets the extent
to win (its left subtree is complete) and folds win into the window below.  Model leaves return hits that ignore the extent (a mesh:
shape/src/blas.rs:468), respect it (an analytic shape) or lie in front of their own box (a ParallelQuad's mirrored quadrants), with
ties among them: both walks must test the same boxes against the same extents in the same order, enter the same leaves with the same
extents, and keep the same hit.  No GPU needed (the GPU side: tests/test_gpu_fuzz.py)."""
import math
import random

INF = math.inf


class Node:
    def __init__(self, rnd, depth):
        self.t_low = rnd.choice([0.0, 0.0, rnd.uniform(0, 10)])  # where the ray enters the box
        if depth == 0 or rnd.random() < 0.25:
            self.kids = None
            self.kind = rnd.choice(["mesh", "analytic", "miss"])
            self.t = rnd.choice([1.0, 2.0, 3.0, 5.0, 8.0]) if rnd.random() < 0.5 else rnd.uniform(0, 12)  # (a few values: ties)
            self.name = id(self)
        else:
            self.kids = (Node(rnd, depth - 1), Node(rnd, depth - 1))

    def leaf_hit(self, extent):
        if self.kind == "miss":
            return None
        if self.kind == "analytic" and not self.t < extent:  # truncated_t
            return None
        return self.t  # a mesh's hit may lie beyond the extent; any hit may lie in front of t_low (the box is only a filter)


def recursive(node, ray, log):  # ray = [t_max]
    log.append(("box", id(node), ray[0]))
    if not node.t_low <= ray[0]:
        return None
    if node.kids is None:
        log.append(("leaf", node.name, ray[0]))
        t = node.leaf_hit(ray[0])
        return None if t is None else (t, node.name)
    left = recursive(node.kids[0], ray, log)
    if left is not None:
        ray[0] = left[0]
    right = recursive(node.kids[1], ray, log)
    if left is None or right is None:
        return left if right is None else right
    return left if left[0] < right[0] else right


def stack_walk(root, t_max, log):
    stack = [(root, False, None, None)]  # (node, is a right sibling, saved window value, saved window flag)
    best = None
    win, win_has = 0.0, False
    while stack:
        node, sibling, below, below_has = stack.pop()
        if sibling:  # the left subtree has returned
            if win_has:
                t_max = win
            if below_has and (not win_has or below < win):
                win = below
            win_has = win_has or below_has
        log.append(("box", id(node), t_max))
        if not node.t_low <= t_max:
            continue
        if node.kids is not None:
            stack.append((node.kids[1], True, win, win_has))
            stack.append((node.kids[0], False, None, None))
            win_has = False
            continue
        log.append(("leaf", node.name, t_max))
        t = node.leaf_hit(t_max)
        if t is not None:
            if not win_has or not win < t:
                win = t
            win_has = True
            if best is None or not best[0] < t:
                best = (t, node.name)
    return best


def test_the_window_stack_is_the_recursion():
    rnd = random.Random(7)
    rises = differing_from_best_so_far = 0
    for case in range(4000):
        root = Node(rnd, rnd.randint(1, 6))
        t0 = rnd.choice([INF, INF, rnd.uniform(1, 12)])
        log_r, log_s = [], []
        want = recursive(root, [t0], log_r)
        got = stack_walk(root, t0, log_s)
        assert log_r == log_s, case
        assert want == got, case
        extents = [e for what, _, e in log_r if what == "box"]
        rises += any(b > a for a, b in zip(extents, extents[1:]))
        # what rounds 1-3 walked with: the extent at the best hit so far — not the same walk once the extent has risen
        best, t_max, simple = None, t0, []
        stack = [root]
        while stack:
            node = stack.pop()
            simple.append(("box", id(node), t_max))
            if not node.t_low <= t_max:
                continue
            if node.kids is not None:
                stack += [node.kids[1], node.kids[0]]
                continue
            t = node.leaf_hit(t_max)
            if t is not None and (best is None or not best[0] < t):
                best, t_max = (t, node.name), t
        differing_from_best_so_far += best != want
    assert rises > 500  # the cases do exercise a rising extent ...
    assert 0 < differing_from_best_so_far < rises  # ... and it decides the result in some of them (hits in front of their box, ties)
