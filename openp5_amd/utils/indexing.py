"""Item-ID assignment (sequential / random / collaborative) with the on-disk caches the reference writes next to
`user_sequence.txt`.  Behaviour follows /root/reference/src/src_t5/utils/indexing.py:11-333:
  * users are numbered 1.. in file order, items get string ids;
  * sequential: ids "1001", "1002", ... in first-seen order over every user's training prefix (all but the last two
    interactions), then over the held-out last two (:49-58);
  * random: a shuffled assignment of the same id range (:97-106);
  * collaborative: recursive spectral clustering of the training co-occurrence graph, ids are `<CIa><CIb>...`
    token strings (:149-256)."""
import os
import random
from collections import defaultdict
from itertools import combinations

import numpy as np

from . import utils


def get_dict_from_lines(lines):
    out = {}
    for line in lines:
        k, v = line.split(" ")[:2]
        out[k] = v
    return out


def construct_user_sequence_dict(user_sequence):
    out = {}
    for line in user_sequence:
        parts = line.split(" ")
        out[parts[0]] = parts[1:]
    return out


def generate_user_map(user_sequence_dict):
    return {user: str(i + 1) for i, user in enumerate(user_sequence_dict.keys())}


def reindex(user_sequence_dict, user_map, item_map):
    return {user_map[u]: [item_map[i] for i in items] for u, items in user_sequence_dict.items()}


def _cached(data_path, dataset, item_file, seq_file, user_sequence_dict, make_item_map):
    folder = os.path.join(data_path, dataset)
    user_index_file = os.path.join(folder, "user_indexing.txt")
    item_index_file = os.path.join(folder, item_file)
    reindex_sequence_file = os.path.join(folder, seq_file)
    if os.path.exists(reindex_sequence_file):
        return (construct_user_sequence_dict(utils.ReadLineFromFile(reindex_sequence_file)),
                get_dict_from_lines(utils.ReadLineFromFile(item_index_file)))
    if os.path.exists(user_index_file):
        user_map = get_dict_from_lines(utils.ReadLineFromFile(user_index_file))
    else:
        user_map = generate_user_map(user_sequence_dict)
        utils.WriteDictToFile(user_index_file, user_map)
    if os.path.exists(item_index_file):
        item_map = get_dict_from_lines(utils.ReadLineFromFile(item_index_file))
    else:
        item_map = make_item_map()
        utils.WriteDictToFile(item_index_file, item_map)
    out = reindex(user_sequence_dict, user_map, item_map)
    utils.WriteDictToFile(reindex_sequence_file, out)
    return out, item_map


def sequential_indexing(data_path, dataset, user_sequence_dict, order):
    def make():
        if order == "original":
            users = list(user_sequence_dict.keys())
        else:
            users = sorted(user_sequence_dict, key=lambda u: len(user_sequence_dict[u]), reverse=(order == "long2short"))
        item_map = {}
        for part in (slice(None, -2), slice(-2, None)):
            for u in users:
                for item in user_sequence_dict[u][part]:
                    if item not in item_map:
                        item_map[item] = str(len(item_map) + 1001)
        return item_map
    return _cached(data_path, dataset, f"item_sequential_indexing_{order}.txt", f"user_sequence_sequential_indexing_{order}.txt",
                   user_sequence_dict, make)


def random_indexing(data_path, dataset, user_sequence_dict):
    def make():
        items = set()
        for seq in user_sequence_dict.values():
            items.update(seq)
        items = list(items)
        random.shuffle(items)
        return {item: str(i + 1001) for i, item in enumerate(items)}
    return _cached(data_path, dataset, "item_random_indexing.txt", "user_sequence_random_indexing.txt", user_sequence_dict, make)


def collaborative_indexing(data_path, dataset, user_sequence_dict, token_size, cluster_num, last_token, float32):
    return _cached(data_path, dataset, f"item_collaborative_indexing_{token_size}_{cluster_num}_{last_token}.txt",
                   f"user_sequence_collaborative_indexing_{token_size}_{cluster_num}_{last_token}.txt", user_sequence_dict,
                   lambda: generate_collaborative_id(user_sequence_dict, token_size, cluster_num, last_token, float32))


def _spectral_labels(adj, cluster_num):
    from sklearn.cluster import SpectralClustering
    return SpectralClustering(n_clusters=cluster_num, assign_labels="cluster_qr", random_state=0, affinity="precomputed").fit(adj).labels_.tolist()


def add_token_to_indexing(item_map, grouping, index_now, token_size):
    for group in grouping:
        index_now = index_now % token_size
        for item, _ in grouping[group]:
            item_map[item] = item_map.get(item, "") + f"<CI{index_now}>"
        index_now += 1
    return item_map, index_now


def add_last_token_to_indexing_random(item_map, item_list, token_size):
    last = random.sample(range(token_size), len(item_list))
    for item, t in zip(item_list, last):
        item_map[item] = item_map.get(item, "") + f"<CI{t}>"
    return item_map


def add_last_token_to_indexing_sequential(item_map, item_list, token_size):
    for i, item in enumerate(item_list):
        item_map[item] = item_map.get(item, "") + f"<CI{i}>"
    return item_map


def generate_collaborative_id(user_sequence_dict, token_size, cluster_num, last_token, float32):
    all_items, train_items = set(), set()
    for seq in user_sequence_dict.values():
        all_items.update(seq)
        train_items.update(seq[:-2])
    item2id = {item: i for i, item in enumerate(train_items)}
    id2item = {i: item for item, i in item2id.items()}
    adj = np.zeros((len(item2id), len(item2id)), dtype=np.float32 if float32 > 0 else np.float64)
    for seq in user_sequence_dict.values():
        for a, b in combinations(seq[:-2], 2):
            adj[item2id[a]][item2id[b]] += 1
            adj[item2id[b]][item2id[a]] += 1
    add_last = add_last_token_to_indexing_sequential if last_token == "sequential" else add_last_token_to_indexing_random

    def group_by(labels, members):
        grouping = defaultdict(list)
        for lab, m in zip(labels, members):
            grouping[lab].append(m)
        return grouping

    grouping = group_by(_spectral_labels(adj, cluster_num), [(id2item[i], i) for i in range(len(id2item))])
    item_map, index_now = add_token_to_indexing({}, grouping, 0, token_size)
    queue = list(grouping.values())
    while queue:
        members = queue.pop(0)
        if len(members) <= token_size:
            item_map = add_last(item_map, [m[0] for m in members], token_size)
            continue
        idx = [m[1] for m in members]
        sub = adj[np.ix_(idx, idx)].copy()
        np.fill_diagonal(sub, 0)
        grouping = group_by(_spectral_labels(sub, cluster_num), members)
        item_map, index_now = add_token_to_indexing(item_map, grouping, index_now, token_size)
        queue.extend(grouping.values())
    remaining = list(all_items - train_items)
    if remaining:
        item_map = add_last(item_map, remaining, token_size)
    return item_map
