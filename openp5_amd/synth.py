"""Synthetic stand-ins for the datasets the reference downloads (data/*/README.md ship links only; there is no network
here): `user_sequence.txt` files with the user / item / interaction counts of README.md:27-37 (5-core, time-ordered,
Zipf item popularity, log-normal sequence lengths), and a small prompt-template file in the reference's format."""
import os

import numpy as np

STATS = {  # users, items, interactions  (/root/reference/README.md:27-37)
    "ML100K": (943, 1349, 99287),
    "ML1M": (6040, 3416, 999611),
    "Beauty": (22363, 12101, 198502),
    "Yelp": (277631, 112394, 4250483),
    "Toy": (30, 40, 260),
}

PROMPTS = """sequential; seen; Considering {dataset} user_{user_id} has interacted with {dataset} items {history} . What is the next recommendation for the user ?; {dataset} {target}
sequential; seen; {dataset} user_{user_id} has interacted with {dataset} items {history} , predict the next item for the user ?; {dataset} {target}
sequential; seen; What would {dataset} user_{user_id} be likely to choose next after {dataset} items {history} ?; {dataset} {target}
sequential; unseen; Which item should {dataset} user_{user_id} see next after {dataset} items {history} ?; {dataset} {target}
straightforward; seen; What should we recommend for {dataset} user_{user_id} ?; {dataset} {target}
straightforward; seen; Do you have any suggested items for {dataset} user_{user_id} ?; {dataset} {target}
straightforward; unseen; Which item fits {dataset} user_{user_id} ?; {dataset} {target}
"""


def write_prompt_file(path):
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    with open(path, "w") as f:
        f.write(PROMPTS)
    return path


def make_user_sequences(dataset, seed=2023, n_users=None, n_items=None, n_inter=None, signal=None):
    """`signal="chain"`: a LEARNABLE stand-in -- every user walks one fixed random cycle over the items from a random start, with a
    jump to a random item every ~12 steps: the next item is a function of the last one, so a model trained on it becomes confident
    (tests/test_gpu_dataset.py needs decision margins well above the bf16 score error).  Default: Zipf popularity, no structure."""
    u, i, n = STATS[dataset]
    n_users, n_items, n_inter = n_users or u, n_items or i, n_inter or n
    rng = np.random.default_rng(seed)
    if signal == "chain":
        succ = rng.permutation(n_items)                       # one cycle-ish successor map
        nxt = np.empty(n_items, dtype=np.int64)
        nxt[succ] = np.roll(succ, -1)
        k = max(6, min(n_items // 3, int(round(n_inter / n_users))))
        lines = []
        for uid in range(n_users):
            cur, seq, seen = int(rng.integers(n_items)), [], set()
            while len(seq) < k:
                if cur in seen or rng.random() < 1.0 / 12.0:
                    cand = [x for x in rng.permutation(n_items)[:8] if int(x) not in seen]
                    if not cand:
                        break
                    cur = int(cand[0])
                seq.append(cur)
                seen.add(cur)
                cur = int(nxt[cur])
            lines.append(f"U{uid} " + " ".join(f"I{it}" for it in seq))
        return lines
    lens = 5 + rng.lognormal(mean=np.log(max(2.0, n_inter / n_users - 5)) - 0.5, sigma=1.0, size=n_users)
    lens = np.clip(np.round(lens * (n_inter / lens.sum())), 5, n_items).astype(int)
    pop = 1.0 / np.arange(1, n_items + 1) ** 1.0
    pop /= pop.sum()
    lines = []
    for uid in range(n_users):
        k = int(lens[uid])
        items = rng.choice(n_items, size=min(k, n_items), replace=False, p=pop)
        lines.append(f"U{uid} " + " ".join(f"I{it}" for it in items))
    return lines


def write_dataset(data_path, dataset, **kw):
    folder = os.path.join(data_path, dataset)
    os.makedirs(folder, exist_ok=True)
    path = os.path.join(folder, "user_sequence.txt")
    if not os.path.exists(path):
        with open(path, "w") as f:
            f.write("\n".join(make_user_sequences(dataset, **kw)) + "\n")
    return path
