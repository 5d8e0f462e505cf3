#!/usr/bin/env python3
"""The argparse Namespace the REFERENCE's own `opts.parse_opt()` (P/opts.py:6-179) produces, stored as data (JSON):

  opts_default.json      -- no command line at all (caption_model = 'transformer', rnn_size = 1300, use_bn = 1, use_box = 1 ...)
  opts_topdown512.json   -- `--caption_model topdown --rnn_size 512`, the two flags that select the hot path of BASELINE
                            configs[1] ("TopDown attention LSTM ... hidden 512"); every other flag at the reference default

Build-container only (imports /root/reference/pivot_based_eccv2018/opts.py).  `--checkpoint_path save/fixture` keeps
parse_opt from stamping the current time into `id` / `checkpoint_path`.

    python tests/golden/make_golden_opts.py
"""
import importlib.util
import json
import os
import sys

os.environ["PYTHONDONTWRITEBYTECODE"] = "1"
sys.dont_write_bytecode = True
P = "/root/reference/pivot_based_eccv2018"
HERE = os.path.dirname(os.path.abspath(__file__))

spec = importlib.util.spec_from_file_location("refopts", os.path.join(P, "opts.py"))
refopts = importlib.util.module_from_spec(spec)
spec.loader.exec_module(refopts)


def dump(name, argv):
    sys.argv = ["train.py", "--checkpoint_path", "save/fixture"] + argv
    ns = refopts.parse_opt()
    path = os.path.join(HERE, name + ".json")
    with open(path, "w") as f:
        json.dump(vars(ns), f, indent=1, sort_keys=True)
    print("wrote %s (%d flags)" % (path, len(vars(ns))))


if __name__ == "__main__":
    dump("opts_default", [])
    dump("opts_topdown512", ["--caption_model", "topdown", "--rnn_size", "512"])
