"""Item-ID prefix trie (host side) and its CSR compilation for the device beam search.

`Trie` mirrors /root/reference/src/src_t5/utils/generation_trie.py:7-88 (same constructor, `add`, `get`,
`__len__`, `__iter__`, `trie_dict`; the `append_trie` hook of the reference is kept and compiled for the device too).  `prefix_allowed_tokens_fn`
mirrors generation_trie.py:91-97 and additionally exposes `.candidate_trie`, which `P5T5Native.generate` uses to
run the constraint on the device instead of calling back into Python per (batch x beam) row per step.
"""
from typing import Dict, Iterable, List, Optional, Sequence

import numpy as np


class Trie(object):
    def __init__(self, sequences: Optional[Iterable[Sequence[int]]] = None):
        self.trie_dict: Dict[int, dict] = {}
        self.len = 0
        if sequences:
            for sequence in sequences:
                self.add(sequence)
        self.append_trie = None
        self.bos_token_id = None

    def append(self, trie, bos_token_id):
        self.append_trie = trie
        self.bos_token_id = bos_token_id

    def add(self, sequence: Sequence[int]):
        node = self.trie_dict
        for tok in sequence:
            node = node.setdefault(int(tok), {})
        self.len += 1

    def get(self, prefix_sequence: Sequence[int]) -> List[int]:
        node = self.trie_dict
        for i, tok in enumerate(prefix_sequence):
            if tok in node:
                node = node[tok]
            elif self.append_trie is not None:
                return self.append_trie.get(list(prefix_sequence[i:]))
            else:
                return []
        out = list(node.keys())
        if self.append_trie is not None and self.bos_token_id in out:
            out.remove(self.bos_token_id)
            out += list(self.append_trie.trie_dict.keys())
        return out

    @staticmethod
    def load_from_dict(trie_dict):
        trie = Trie()
        trie.trie_dict = trie_dict
        trie.len = sum(1 for _ in trie)
        return trie

    def __iter__(self):
        def _traverse(prefix, node):
            if node:
                for tok in node:
                    yield from _traverse(prefix + [tok], node[tok])
            else:
                yield prefix
        return _traverse([], self.trie_dict)

    def __len__(self):
        return self.len

    def __getitem__(self, value):
        return self.get(value)


def prefix_allowed_tokens_fn(candidate_trie):
    def prefix_allowed_tokens(batch_id, sentence):
        sentence = sentence.tolist() if hasattr(sentence, "tolist") else list(sentence)
        return candidate_trie.get(sentence)

    prefix_allowed_tokens.candidate_trie = candidate_trie
    return prefix_allowed_tokens


def exact_match(predictions, targets, k):
    """generation_trie.py:99-109."""
    correct = 0
    for b, t in enumerate(targets):
        if t in predictions[b * k:(b + 1) * k]:
            correct += 1
    return correct


class CompiledTrie:
    """CSR form: node 0 is the empty prefix; children of node n are edges child_off[n]..child_off[n+1] with token
    `child_tok` (sorted ascending, matching HF's flat beam*V+token tie order) leading to node `child_node`."""

    def __init__(self, child_off: np.ndarray, child_tok: np.ndarray, child_node: np.ndarray):
        self.child_off = np.ascontiguousarray(child_off, dtype=np.int32)
        self.child_tok = np.ascontiguousarray(child_tok, dtype=np.int32)
        self.child_node = np.ascontiguousarray(child_node, dtype=np.int32)
        self.n_nodes = len(self.child_off) - 1
        self.max_children = int(np.max(np.diff(self.child_off))) if self.n_nodes > 0 else 0
        self._dev = {}
        self._max_depth = None
        self.grafted = False            # compiled from a trie with an appended trie: a DAG, item bookkeeping does not apply

    @property
    def max_depth(self) -> int:
        """Number of tokens of the longest root-to-leaf path (including the decoder start token): no hypothesis of a search
        constrained by this trie is longer, which bounds the number of decode steps worth enqueuing."""
        if self._max_depth is None:
            # level-by-level frontier expansion from the root (node 0): no assumption about how the nodes are numbered, and one
            # vectorised gather per trie level instead of a Python loop over every edge
            depth, frontier = 0, np.zeros(1, dtype=np.int64)
            off = self.child_off.astype(np.int64)
            while self.n_nodes > 0 and frontier.size:
                lo, cnt = off[frontier], off[frontier + 1] - off[frontier]
                total = int(cnt.sum())
                if total == 0:
                    break
                starts = np.repeat(lo - np.concatenate(([0], np.cumsum(cnt)[:-1])), cnt)
                # (unique: a grafted trie is a DAG whose sub-trie is shared by every graft point -- the frontier must count nodes, not paths)
                frontier = np.unique(self.child_node[starts + np.arange(total, dtype=np.int64)].astype(np.int64))
                depth += 1
                assert depth <= self.n_nodes, "CompiledTrie: the child table has a cycle"
            self._max_depth = depth
        return self._max_depth

    def forced_prefix(self, start_id: int, eos_id: int, limit: int = 16):
        """(tokens, nodes) of the chain every item shares behind the decoder start token: from the start node on, as long as a node has
        exactly ONE child and that child is not </s>.  OpenP5's item ids all begin with "<dataset> item _" (the `{dataset} {target}` target
        template), so the first steps of every beam search under the item trie are forced -- p5_generate_set_forced_prefix turns them into
        one teacher-forced pass.  Cached per (start, eos)."""
        key = (int(start_id), int(eos_id), int(limit))
        cache = self.__dict__.setdefault("_forced", {})
        if key not in cache:
            toks, nodes = [], []
            node = -1
            for c in range(int(self.child_off[0]), int(self.child_off[1])):
                if int(self.child_tok[c]) == start_id:
                    node = int(self.child_node[c])
            while node >= 0 and len(toks) < limit:
                lo, hi = int(self.child_off[node]), int(self.child_off[node + 1])
                if hi - lo != 1 or int(self.child_tok[lo]) == eos_id:
                    break
                toks.append(int(self.child_tok[lo]))
                node = int(self.child_node[lo])
                nodes.append(node)
            cache[key] = (toks, nodes)
        return cache[key]

    @staticmethod
    def from_dict(trie_dict: Dict[int, dict]) -> "CompiledTrie":
        off, tok, nxt = [0], [], []
        # breadth-first so that siblings are contiguous
        queue = [trie_dict]
        next_id = 1
        qi = 0
        while qi < len(queue):
            node = queue[qi]
            qi += 1
            for t in sorted(node.keys()):
                tok.append(int(t))
                nxt.append(next_id)
                next_id += 1
                queue.append(node[t])
            off.append(len(tok))
        return CompiledTrie(np.asarray(off), np.asarray(tok), np.asarray(nxt))

    @staticmethod
    def from_trie(trie) -> "CompiledTrie":
        """Compile a `Trie` (ours or the reference's: anything with `.trie_dict`, optionally `.append_trie` / `.bos_token_id`).
        An appended trie (generation_trie.py:19-21, 47-70) is GRAFTED at compile time, so the device search needs nothing new:
        wherever a node of the main trie has a `bos_token_id` child, the allowed tokens are its other children plus the appended
        trie's root tokens (`output.remove(bos); output += append_trie.trie_dict.keys()`, :54-57), and a token the main node does
        not have leads into the appended trie (`append_trie.get(prefix_sequence)`, :66-68) -- the main trie wins when both have
        the token (`elif prefix_sequence[0] in trie_dict`, :59).  The result is a DAG in the same CSR form (the appended trie's
        nodes are shared by every graft point), children sorted by token as everywhere else."""
        app = getattr(trie, "append_trie", None)
        main = CompiledTrie.from_dict(trie.trie_dict)
        if app is None:
            return main
        bos = trie.bos_token_id
        sub = CompiledTrie.from_trie(app)               # (the appended trie may itself carry one)
        base = main.n_nodes
        rt, rn = sub.children(0)
        off, tok, nxt = [0], [], []
        for n in range(main.n_nodes):
            t, c = main.children(n)
            edges = [(int(a), int(b)) for a, b in zip(t, c)]
            have = {a for a, _ in edges}
            if bos in have:
                edges = [(a, b) for a, b in edges if a != bos]
                edges += [(int(a), base + int(b)) for a, b in zip(rt, rn) if int(a) not in have]
                if bos in {int(a) for a in rt}:         # bos itself stays allowed then, and the main trie's edge is the one taken
                    edges.append((int(bos), int(c[list(t).index(bos)])))
            for a, b in sorted(edges):
                tok.append(a)
                nxt.append(b)
            off.append(len(tok))
        for n in range(sub.n_nodes):
            t, c = sub.children(n)
            for a, b in zip(t, c):
                tok.append(int(a))
                nxt.append(base + int(b))
            off.append(len(tok))
        out = CompiledTrie(np.asarray(off), np.asarray(tok), np.asarray(nxt))
        out.grafted = True
        return out

    @staticmethod
    def from_sequences(seqs: Iterable[Sequence[int]]) -> "CompiledTrie":
        return CompiledTrie.from_trie(Trie(seqs))

    # ---- per-user exclusion without per-user tries (DistributedRunner.py:286-297 rebuilds a Trie over
    # `all_items - positive` for every user; the same trie is "the full trie minus every node all of whose items are
    # excluded", which is a bitmap over node ids) ----
    def index_items(self, seqs: Sequence[Sequence[int]]) -> None:
        """Record, for each item sequence (in the caller's order), the node ids along its path."""
        if self.grafted:
            raise NotImplementedError("per-item bookkeeping (history exclusion) on a trie with an appended trie")
        edge = {}
        off, tok, nxt = self.child_off, self.child_tok, self.child_node
        for n in range(self.n_nodes):
            for e in range(off[n], off[n + 1]):
                edge[(n, int(tok[e]))] = int(nxt[e])
        depth = max((len(q) for q in seqs), default=0)
        paths = np.full((len(seqs), depth), -1, dtype=np.int32)
        for i, q in enumerate(seqs):
            n = 0
            for j, t in enumerate(q):
                n = edge[(n, int(t))]
                paths[i, j] = n
        self.item_paths = paths
        cnt = np.zeros(self.n_nodes, dtype=np.int64)
        np.add.at(cnt, paths[paths >= 0], 1)
        self.items_under = cnt
        self.excluded_words = (self.n_nodes + 31) // 32

    def excluded_bitmap(self, excluded_items: Sequence[Sequence[int]]) -> np.ndarray:
        """uint32 [B, excluded_words]: bit n of row b is set when every item below node n is in excluded_items[b]
        (indices into the list given to `index_items`)."""
        bm = np.zeros((len(excluded_items), self.excluded_words), dtype=np.uint32)
        for b, items in enumerate(excluded_items):
            items = np.unique(np.asarray(list(items), dtype=np.int64))      # a duplicate must not count twice
            if items.size == 0:
                continue
            nodes = self.item_paths[items].ravel()
            u, c = np.unique(nodes[nodes >= 0], return_counts=True)
            dead = u[self.items_under[u] == c]
            np.bitwise_or.at(bm[b], dead >> 5, (np.uint32(1) << (dead & 31).astype(np.uint32)))
        return bm

    def children(self, node: int):
        a, b = self.child_off[node], self.child_off[node + 1]
        return self.child_tok[a:b], self.child_node[a:b]

    def device_arrays(self, device):
        import torch
        key = str(device)
        if key not in self._dev:
            self._dev[key] = tuple(torch.from_numpy(a).to(device) for a in (self.child_off, self.child_tok, self.child_node))
        return self._dev[key]


def find_trie(fn):
    """Recover the Trie behind a prefix_allowed_tokens_fn: ours (`.candidate_trie`) or the reference's closure
    (generation_trie.py:91-97 closes over `candidate_trie`)."""
    t = getattr(fn, "candidate_trie", None)
    if t is not None:
        return t
    for cell in (getattr(fn, "__closure__", None) or ()):
        try:
            obj = cell.cell_contents
        except ValueError:
            continue
        if hasattr(obj, "trie_dict") and hasattr(obj, "get"):
            return obj
    return None
