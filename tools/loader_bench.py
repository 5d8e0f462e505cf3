#!/usr/bin/env python3
"""Input pipeline measurement: the batch-assembly kernel against its HBM roofline, the loader end
to end from files, and the numpy oracle (= the reference's per-image work) on the host cores beside it.

    python tools/loader_bench.py [--images 128] [--regions 36] [--feat 2048] [--batches 20]
"""
import argparse
import json
import os
import random
import shutil
import sys
import tempfile
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--images", type=int, default=128)
    ap.add_argument("--regions", type=int, default=36)
    ap.add_argument("--feat", type=int, default=2048)
    ap.add_argument("--batches", type=int, default=20)
    ap.add_argument("--dataset-images", type=int, default=512)
    ap.add_argument("--read-threads", type=int, default=0, help="reader team size (0: the loader's default)")
    ap.add_argument("--compressed", action="store_true", help="att files as np.savez_compressed writes them (make_bu_data.py:55)")
    a = ap.parse_args()
    from unpaired_image_captioning_amd import _lib
    from unpaired_image_captioning_amd._lib import check, ptr, stream
    from unpaired_image_captioning_amd.misc.dataloader.dataloader import DataLoader, padded_width
    from dataset_files import loader_opt, write_dataset
    from oracle import dataloader as O
    lib = _lib.load()
    n, R, D = a.images, a.regions, a.feat
    Dout, ld = D + 5, padded_width(D + 5)
    rng = np.random.default_rng(0)

    # ---- 1. the kernel alone: fixed 36 regions per image, boxes, both norms (the reference's defaults) ----
    sets = []
    n_sets = max(2, int(np.ceil(300e6 / (n * R * (D + ld) * 4))) + 1)          # rotate through > 256 MiB: HBM-cold
    start = torch.arange(0, (n + 1) * R, R, dtype=torch.int32, device="cuda")
    slot = torch.arange(n, dtype=torch.int32, device="cuda")
    hw = torch.tensor([[480., 640., 480. * 640.]] * n, device="cuda")
    for _ in range(n_sets):
        feat = torch.rand(n * R, D, device="cuda")
        xy = torch.rand(n * R, 2, device="cuda") * 200
        box = torch.cat([xy, xy + 10 + torch.rand(n * R, 2, device="cuda") * 200], 1).contiguous()
        sets.append((feat, box, torch.empty(n, R, ld, device="cuda"), torch.empty(n, R, device="cuda")))

    def launch(s):
        feat, box, out, m = s
        check(lib.uic_att_batch_assemble(ptr(feat), ptr(box), ptr(start), ptr(hw), ptr(slot), n, D, 1, 1, R, ld, ptr(out), ptr(m),
                                         stream()), "assemble")
    for s in sets:
        launch(s)
    torch.cuda.synchronize()

    def timed(pick, reps=200):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(reps):
            launch(pick(i))
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) * 1e3 / reps
    us_resident = timed(lambda i: sets[0])
    us_cold = timed(lambda i: sets[i % n_sets])
    algo = n * R * (D * 4 + 16) + n * R * ld * 4 + n * R * 4                   # read features + boxes, write rows + masks

    # ---- 2. end to end from files, and the oracle on the same files ----
    tmp = tempfile.mkdtemp(prefix="uic_loader_bench_", dir="/tmp")
    try:
        N = a.dataset_images
        att = [np.abs(rng.standard_normal((R, D))).astype(np.float32) for _ in range(N)]
        box = []
        for _ in range(N):
            xy = rng.uniform(0, 200, (R, 2))
            box.append(np.hstack([xy, xy + rng.uniform(10, 200, (R, 2))]).astype(np.float32))
        fc = [x.mean(0) for x in att]
        labels = rng.integers(1, 9487, (N * 5, 16)).astype(np.uint32)
        ends = np.arange(5, N * 5 + 1, 5)
        label_path = write_dataset(tmp, att, box, fc, [(480, 640)] * N, list(range(N)), labels, ends - 4, ends, 9487,
                                   label_format="npz")
        if a.compressed:
            for i in range(N):
                np.savez_compressed(os.path.join(tmp, "att", "%d.npz" % i), feat=att[i])
        opt = loader_opt(tmp, label_path, n, 5, D, Dout, 1, 1, 1)
        loader = DataLoader(opt, read_threads=a.read_threads or None)
        random.seed(0)
        for _ in range(3):
            loader.get_batch("train")
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        per_batch = []
        for _ in range(a.batches):
            t1 = time.perf_counter()
            d = loader.get_batch("train")
            per_batch.append(round((time.perf_counter() - t1) * 1e3, 2))
        torch.cuda.synchronize()
        dt_loader = (time.perf_counter() - t0) / a.batches
        print("get_batch, ms per call: " + " ".join("%.2f" % x for x in per_batch), file=sys.stderr)

        # ---- 3. training from files: loader -> Trainer.train, the reference's defaults (use_box: 2053 features, use_bn 1) ----
        import argparse as _ap
        from unpaired_image_captioning_amd.trainer import Trainer
        topt = _ap.Namespace(vocab_size=9487, input_encoding_size=512, rnn_size=512, num_layers=1, drop_prob_lm=0.5,
                             seq_length=16, fc_feat_size=D, att_feat_size=Dout, att_hid_size=512, use_bn=1, logit_layers=1,
                             caption_model="topdown", compute_dtype="bf16", seed=1, i2t_learning_rate=5e-4, i2t_train_flag=1,
                             seq_per_img=5)
        tr = Trainer(topt)
        tr.i2t_model.cuda()
        tr.build_optimizer()
        fetch = lambda: loader.get_batch("train")
        cur = fetch()
        for _ in range(4):
            tr.train(cur, next_data=fetch)              # the next batch is fetched after this step is enqueued
            cur = tr.next_data
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(a.batches):
            tr.train(cur, next_data=fetch)
            cur = tr.next_data
        torch.cuda.synchronize()
        dt_train = (time.perf_counter() - t0) / a.batches
        syn = {k: (v.clone() if torch.is_tensor(v) else v) for k, v in cur.items()}
        for _ in range(3):
            tr.train(syn, next_data=syn)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(a.batches):
            tr.train(syn, next_data=syn)
        torch.cuda.synchronize()
        dt_resident = (time.perf_counter() - t0) / a.batches

        # the reference's per-image work (numpy) + merge, files read the same way, one process (its DataLoader workers: 4)
        def oracle_batch(first):
            fcs, atts, rows, gts, infos = [], [], [], [], []
            for ix in range(first, first + n):
                ix %= N
                f = np.load(os.path.join(tmp, "fc", "%d.npz" % ix))["feat"]
                x = np.load(os.path.join(tmp, "att", "%d.npz" % ix))["feat"]
                b = np.load(os.path.join(tmp, "box", "%d.npy" % ix))
                atts.append(O.region_features(x, b, 480, 640, 1, 1))
                fcs.append(f)
                rows.append(labels[ix * 5:(ix + 1) * 5].astype("int"))
                gts.append(labels[ix * 5:(ix + 1) * 5])
                infos.append({"ix": ix})
            return O.merge_batch(fcs, atts, np.vstack(rows), gts, infos, 5, 16)
        oracle_batch(0)
        reps = 3
        t0 = time.perf_counter()
        for r in range(reps):
            oracle_batch(r * n)
        dt_oracle = (time.perf_counter() - t0) / reps
        t0 = time.perf_counter()
        for r in range(reps):
            for ix in range(r * n, (r + 1) * n):
                ix %= N
                np.load(os.path.join(tmp, "fc", "%d.npz" % ix))["feat"]
                np.load(os.path.join(tmp, "att", "%d.npz" % ix))["feat"]
                np.load(os.path.join(tmp, "box", "%d.npy" % ix))
        dt_files = (time.perf_counter() - t0) / reps
    finally:
        shutil.rmtree(tmp)

    print(json.dumps({
        "what": "input pipeline, %d images x %d regions x %d (+5 box) features, norm_att_feat = norm_box_feat = use_box = 1" % (n, R, D),
        "kernel": {"name": "att_batch_assemble_kernel", "algorithmic_MB": round(algo / 1e6, 2),
                   "us_per_launch_resident": round(us_resident, 2), "us_per_launch_hbm_cold": round(us_cold, 2),
                   "GBps_resident": round(algo / us_resident / 1e3, 1), "GBps_hbm_cold": round(algo / us_cold / 1e3, 1),
                   "frac_of_8TBps_hbm_cold": round(algo / us_cold / 1e3 / 8000, 3), "rotating_sets": n_sets},
        "loader_end_to_end": {"ms_per_batch": round(dt_loader * 1e3, 2), "images_per_s": round(n / dt_loader, 1),
                              "reader_threads": loader.read_threads_deflate if a.compressed else loader.read_threads, "att_files": "deflated" if a.compressed else "stored",
                              "note": "files in the page cache, library reader team straight into pinned staging, read-ahead of the next batch, un-replicated H2D"},
        "train_from_files": {"ms_per_step": round(dt_train * 1e3, 2), "captions_per_s": round(n * 5 / dt_train, 0),
                             "ms_per_step_same_batch_resident": round(dt_resident * 1e3, 2),
                             "note": "Trainer.train(loader.get_batch(..)) loop, TopDown 512/512/9488, bf16, use_bn 1, 640 caption rows per step"},
        "cpu_oracle": {"ms_per_batch": round(dt_oracle * 1e3, 2), "images_per_s": round(n / dt_oracle, 1),
                       "of_which_file_reads_ms": round(dt_files * 1e3, 2), "cores": 1,
                       "note": "numpy restatement of the reference's __getitem__ + get_batch merge (S = 5 replication included)"},
    }))


if __name__ == "__main__":
    main()
