"""Prompt-template file parsing: `task; seen|unseen; input template; output template` per line
(format of /root/reference/prompt.txt; API of /root/reference/src/src_t5/utils/prompt.py:5-60)."""
import os
import re


def _read_lines(path):
    if not os.path.exists(path):
        raise FileNotFoundError(path)
    with open(path, "r") as fd:
        return [line.rstrip("\n") for line in fd]


def load_prompt_template(path, task_list):
    """-> {task: {'seen'|'unseen': {'0': {'Input': ..., 'Output': ...}, '1': ...}}} for the requested tasks only."""
    templates = {}
    for line in _read_lines(path):
        fields = [f.strip() for f in line.split(";")]
        if len(fields) < 4 or fields[0] not in task_list:
            continue
        task, seen = fields[0], fields[1]
        bucket = templates.setdefault(task, {}).setdefault(seen, {})
        bucket[str(len(bucket))] = {"Input": fields[2], "Output": fields[3]}
    return templates


def get_info_from_prompt(prompt_templates):
    """Names of the {placeholders} used by any template."""
    found = set()
    for per_task in prompt_templates.values():
        for per_seen in per_task.values():
            for tpl in per_seen.values():
                found.update(re.findall(r"\{(.*?)\}", tpl["Input"]))
                found.update(re.findall(r"\{(.*?)\}", tpl["Output"]))
    return list(found)


def check_task_prompt(prompt_templates, task_list):
    for task in task_list:
        assert task in prompt_templates, f"No prompt for {task} task"
