"""Minimal rooted-tree object model and Newick I/O for the host side.

The reference keeps cogent3 ``PhyloNode`` objects at its API boundary
(reference: src/sc_supertree/scs.py:7-9, load.py:4).  cogent3 is a third-party
dependency that is not part of the reference tree; this module provides the
duck-typed subset of its node API that the supertree path touches
(SURVEY.md section 2 row 9): child iteration, ``is_tip``, ``name``, ``length``,
``support``, ``get_tip_names``, ``get_sub_tree``, ``get_newick``, ``write``,
``sorted`` and ``same_shape``.  ``construct_supertree`` accepts either these
nodes or real cogent3 nodes (anything exposing the same attributes).

Nothing here runs on the device: tree parsing, restriction and assembly stay on
the host (BASELINE.json north_star).
"""

from __future__ import annotations

import os
from collections.abc import Iterable, Iterator
from pathlib import Path

__all__ = ["TreeNode", "make_tree", "load_tree", "NotCompleted"]


class NotCompleted:
    """Stand-in for ``cogent3.app.composable.NotCompleted``.

    The reference drops such entries from its input list
    (reference: src/sc_supertree/scs.py:82-94).  When cogent3 is importable
    its own class is recognised as well (see ``is_not_completed``).
    """

    def __init__(self, type_: str = "ERROR", origin: str = "", message: str = "") -> None:
        self.type = type_
        self.origin = origin
        self.message = message

    def __bool__(self) -> bool:
        return False

    def __repr__(self) -> str:
        return f"NotCompleted(type={self.type!r}, origin={self.origin!r}, message={self.message!r})"


def is_not_completed(obj: object) -> bool:
    """True for this module's NotCompleted and for cogent3's, when present."""
    if isinstance(obj, NotCompleted):
        return True
    cls = type(obj)
    return cls.__name__ == "NotCompleted" and cls.__module__.startswith("cogent3")


class TreeNode:
    """A rooted tree node: name, branch length, support, ordered children."""

    __slots__ = ("name", "length", "support", "children", "parent")

    def __init__(
        self,
        name: str | None = None,
        children: Iterable["TreeNode"] | None = None,
        length: float | None = None,
        support: float | None = None,
    ) -> None:
        self.name = name
        self.length = length
        self.support = support
        self.parent: TreeNode | None = None
        self.children: list[TreeNode] = []
        if children:
            for c in children:
                self.append(c)

    # -- flat form (what travels between ranks: pickling a linked tree recurses per level) --
    def to_flat(self) -> tuple[list[int], list, list, list]:
        """Preorder ``(parent index, name, length, support)`` lists; iterative, exact."""
        parents: list[int] = []
        names: list = []
        lengths: list = []
        supports: list = []
        stack: list[tuple[TreeNode, int]] = [(self, -1)]
        while stack:
            node, par = stack.pop()
            me = len(parents)
            parents.append(par)
            names.append(node.name)
            lengths.append(node.length)
            supports.append(node.support)
            for child in reversed(node.children):
                stack.append((child, me))
        return parents, names, lengths, supports

    @classmethod
    def from_flat(cls, flat) -> "TreeNode":
        parents, names, lengths, supports = flat
        nodes: list[TreeNode] = []
        for par, name, length, support in zip(parents, names, lengths, supports):
            node = cls(name, None, length, support)
            nodes.append(node)
            if par >= 0:
                nodes[par].append(node)
        return nodes[0]

    def __reduce__(self):
        # (pickle and copy.deepcopy go through the flat form: no recursion per tree level)
        return (TreeNode.from_flat, (self.to_flat(),))

    # -- structure ---------------------------------------------------------
    def append(self, child: "TreeNode") -> None:
        child.parent = self
        self.children.append(child)

    def __iter__(self) -> Iterator["TreeNode"]:
        return iter(self.children)

    def __len__(self) -> int:
        return len(self.children)

    def is_tip(self) -> bool:
        return not self.children

    def is_root(self) -> bool:
        return self.parent is None

    def iter_tips(self) -> Iterator["TreeNode"]:
        stack = [self]
        while stack:
            node = stack.pop()
            if node.children:
                stack.extend(reversed(node.children))
            else:
                yield node

    def iter_nontips(self, include_self: bool = False) -> Iterator["TreeNode"]:
        stack = [self]
        while stack:
            node = stack.pop()
            if node.children:
                if include_self or node is not self:
                    yield node
                stack.extend(reversed(node.children))

    def get_tip_names(self) -> list[str]:
        return [t.name for t in self.iter_tips()]

    def copy(self) -> "TreeNode":
        """Deep copy (iterative, safe for ladder-shaped trees)."""
        root = TreeNode(self.name, None, self.length, self.support)
        stack = [(self, root)]
        while stack:
            src, dst = stack.pop()
            for c in src.children:
                cc = TreeNode(c.name, None, c.length, c.support)
                dst.append(cc)
                stack.append((c, cc))
        return root

    deepcopy = copy

    # -- restriction -------------------------------------------------------
    def get_sub_tree(
        self,
        names: Iterable[str],
        ignore_missing: bool = False,
        as_rooted: bool = True,
    ) -> "TreeNode":
        """Tree induced on ``names``.

        Semantics follow the reference's single call site,
        ``tree.get_sub_tree(names, ignore_missing=True, as_rooted=True)``
        (reference: src/sc_supertree/scs.py:450): tips outside ``names`` are
        dropped, internal nodes left without tips are dropped, internal nodes
        left with one child are spliced out with their branch length added to
        the surviving child, and a root left with one child collapses onto it.
        """
        wanted = set(names)
        if not ignore_missing:
            missing = wanted.difference(self.get_tip_names())
            if missing:
                msg = f"tips not in tree: {sorted(missing)}"
                raise ValueError(msg)
        # post-order, iterative
        result: dict[int, TreeNode | None] = {}
        order: list[TreeNode] = []
        stack = [self]
        while stack:
            node = stack.pop()
            order.append(node)
            stack.extend(node.children)
        for node in reversed(order):
            if not node.children:
                keep = node.name in wanted
                result[id(node)] = (
                    TreeNode(node.name, None, node.length, node.support) if keep else None
                )
                continue
            kept = [result.pop(id(c)) for c in node.children]
            kept = [k for k in kept if k is not None]
            if not kept:
                result[id(node)] = None
            elif len(kept) == 1:
                child = kept[0]
                if node.length is not None and child.length is not None:
                    child.length = node.length + child.length
                child.parent = None
                result[id(node)] = child
            else:
                result[id(node)] = TreeNode(node.name, kept, node.length, node.support)
        out = result[id(self)]
        if out is None:
            msg = "no tips left after restriction"
            raise ValueError(msg)
        out.parent = None
        return out

    # -- comparison --------------------------------------------------------
    def _shape_key(self):
        """Canonical nested-tuple key of the rooted, unordered topology."""
        keys: dict[int, object] = {}
        order: list[TreeNode] = []
        stack = [self]
        while stack:
            node = stack.pop()
            order.append(node)
            stack.extend(node.children)
        for node in reversed(order):
            if not node.children:
                keys[id(node)] = (0, node.name)
            else:
                sub = sorted((keys.pop(id(c)) for c in node.children), key=repr)
                keys[id(node)] = (1, tuple(sub))
        return keys[id(self)]

    def sorted(self) -> "TreeNode":
        """Copy with children ordered canonically (by smallest tip name)."""
        new = self.copy()
        order: list[TreeNode] = []
        stack = [new]
        while stack:
            node = stack.pop()
            order.append(node)
            stack.extend(node.children)
        min_name: dict[int, str] = {}
        for node in reversed(order):
            if not node.children:
                min_name[id(node)] = str(node.name)
            else:
                node.children.sort(key=lambda c: min_name[id(c)])
                min_name[id(node)] = min_name[id(node.children[0])]
        return new

    def same_shape(self, other: "TreeNode") -> bool:
        """Same rooted topology over the same tip names (branch data ignored)."""
        return self._shape_key() == other._shape_key()

    def rooted(self, edge_name: str) -> "TreeNode":
        """A copy re-rooted on the branch above the node named ``edge_name`` (outgroup
        rooting, as cogent3's ``PhyloNode.rooted``): the new root has that node and the rest of
        the tree as its two children; the old root, left with one child, is spliced out."""
        new = self.copy()
        target = None
        stack = [new]
        while stack:
            node = stack.pop()
            if node.name == edge_name and node is not new:
                target = node
                break
            stack.extend(node.children)
        if target is None:
            msg = f"no node named {edge_name!r} below the root"
            raise ValueError(msg)
        if target.parent is new and len(new.children) == 2:
            return new  # already rooted on that branch
        # reverse the parent links on the path from the target's parent up to the old root
        half = None if target.length is None else target.length / 2
        root = TreeNode(None, None, None, None)
        up, up_len = target.parent, half
        up.children.remove(target)
        target.parent = None
        target.length = half
        root.append(target)
        prev = root
        while up is not None:
            nxt, nxt_len = up.parent, up.length
            if nxt is not None:
                nxt.children.remove(up)
            up.parent = None
            up.length = up_len
            prev.append(up)
            prev, up, up_len = up, nxt, nxt_len
        # the old root may be left with a single child: splice it out
        if len(prev.children) == 1 and prev is not root:
            only = prev.children[0]
            parent = prev.parent
            idx = parent.children.index(prev)
            if only.length is not None or prev.length is not None:
                only.length = (only.length or 0.0) + (prev.length or 0.0)
            only.parent = parent
            parent.children[idx] = only
        return root

    # -- Newick ------------------------------------------------------------
    def get_newick(self, with_distances: bool = False, with_node_names: bool = False) -> str:
        pieces: dict[int, str] = {}
        order: list[TreeNode] = []
        stack = [self]
        while stack:
            node = stack.pop()
            order.append(node)
            stack.extend(node.children)
        for node in reversed(order):
            if node.children:
                text = "(" + ",".join(pieces.pop(id(c)) for c in node.children) + ")"
                if with_node_names and node.name and node is not self:
                    text += _quote(node.name)
            else:
                text = _quote(node.name or "")
            if with_distances and node.length is not None and node is not self:
                text += f":{node.length!r}"
            pieces[id(node)] = text
        return pieces[id(self)] + ";"

    def write(self, path: str | os.PathLike, with_distances: bool = True) -> None:
        Path(path).write_text(self.get_newick(with_distances=with_distances) + "\n")

    def __str__(self) -> str:
        return self.get_newick()

    def __repr__(self) -> str:
        return f"TreeNode({self.get_newick()!r})"


_NEEDS_QUOTE = set("()[]':;, \t\n")


def _quote(name: str) -> str:
    if any(ch in _NEEDS_QUOTE for ch in name):
        return "'" + name.replace("'", "''") + "'"
    return name


def _to_number(text: str) -> float | None:
    try:
        return float(text)
    except ValueError:
        return None


def make_tree(newick: str) -> TreeNode:
    """Parse one Newick string into a rooted ``TreeNode`` tree.

    Numeric labels on internal nodes are read as support values, which is
    what the reference's bootstrap weighting consumes through ``node.support``
    (reference: src/sc_supertree/scs.py:563-564,
    tests/test_spectral_cluster_supertree.py:217-219).  Square-bracket
    comments are skipped.  Parsing is iterative.
    """
    s = newick.strip()
    if not s:
        msg = "empty Newick string"
        raise ValueError(msg)
    root = TreeNode()
    cur = root
    i, n = 0, len(s)
    seen_any = False
    # state: after '(' or ',' we start a new node lazily
    pending_new = True  # cur is a fresh node awaiting label/children
    while i < n:
        ch = s[i]
        if ch in " \t\r\n":
            i += 1
        elif ch == "[":
            j = s.find("]", i)
            if j < 0:
                msg = "unterminated comment in Newick string"
                raise ValueError(msg)
            i = j + 1
        elif ch == "(":
            child = TreeNode()
            cur.append(child)
            cur = child
            seen_any = True
            i += 1
        elif ch == ",":
            if cur.parent is None:
                msg = "unbalanced Newick string: ',' at top level"
                raise ValueError(msg)
            sib = TreeNode()
            cur.parent.append(sib)
            cur = sib
            i += 1
        elif ch == ")":
            if cur.parent is None:
                msg = "unbalanced Newick string: too many ')'"
                raise ValueError(msg)
            cur = cur.parent
            i += 1
        elif ch == ";":
            break
        elif ch == ":":
            j = i + 1
            while j < n and s[j] not in ",();[":
                j += 1
            text = s[i + 1 : j].strip()
            cur.length = float(text) if text else None
            i = j
        else:
            if ch == "'":
                j = i + 1
                buf = []
                while j < n:
                    if s[j] == "'":
                        if j + 1 < n and s[j + 1] == "'":
                            buf.append("'")
                            j += 2
                            continue
                        break
                    buf.append(s[j])
                    j += 1
                if j >= n:
                    msg = "unterminated quoted label in Newick string"
                    raise ValueError(msg)
                label = "".join(buf)
                i = j + 1
            else:
                j = i
                while j < n and s[j] not in ",():;[":
                    j += 1
                label = s[i:j].strip()
                i = j
            seen_any = True
            if cur.children:
                num = _to_number(label)
                if num is not None:
                    cur.support = num
                cur.name = label
            else:
                cur.name = label
    if cur is not root:
        msg = "unbalanced Newick string: missing ')'"
        raise ValueError(msg)
    if not seen_any:
        msg = "no tree found in Newick string"
        raise ValueError(msg)
    del pending_new
    return root


def load_tree(path: str | os.PathLike) -> TreeNode:
    """Read a file holding a single Newick tree."""
    return make_tree(Path(path).read_text())


def tip_names_to_tree(tip_names: Iterable[str]) -> TreeNode:
    """Star tree over the names; a single name yields a lone tip.

    Mirrors ``_tip_names_to_tree`` + ``_connect_trees``
    (reference: src/sc_supertree/scs.py:728-746, 390-408).
    """
    tips = [TreeNode(name) for name in tip_names]
    return connect_trees(tips)


def connect_trees(trees: list[TreeNode]) -> TreeNode:
    """Join trees under a new root; one tree is returned unchanged.

    reference: src/sc_supertree/scs.py:390-408
    """
    if len(trees) == 1:
        return trees[0]
    return TreeNode(None, trees)
