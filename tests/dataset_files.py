"""Write a tiny on-disk data set in the reference's formats (per-image att / fc .npz, box .npy, info json, label HDF5 or
.npz) from raw arrays -- shared by the loader tests and tools/loader_bench.py."""
import argparse
import json
import os

import numpy as np


def write_dataset(root, att, box, fc, hw, ids, labels, label_start_ix, label_end_ix, V, splits=None, label_format="h5"):
    for d in ("att", "box", "fc"):
        os.makedirs(os.path.join(root, d), exist_ok=True)
    info = {"images": [], "ix_to_word": {str(i + 1): "w%d" % i for i in range(V)}}
    for i, iid in enumerate(ids):
        np.savez(os.path.join(root, "att", "%d.npz" % iid), feat=att[i])
        if box is not None:
            np.save(os.path.join(root, "box", "%d.npy" % iid), box[i])
        np.savez(os.path.join(root, "fc", "%d.npz" % iid), feat=fc[i])
        info["images"].append({"id": int(iid), "file_path": "img/%d.jpg" % iid, "split": splits[i] if splits else "train",
                               "height": int(hw[i][0]), "width": int(hw[i][1])})
    with open(os.path.join(root, "talk.json"), "w") as f:
        json.dump(info, f)
    arrays = {"labels": np.asarray(labels, dtype=np.uint32), "label_start_ix": np.asarray(label_start_ix, dtype=np.uint32),
              "label_end_ix": np.asarray(label_end_ix, dtype=np.uint32),
              "label_length": (np.asarray(labels) != 0).sum(1).astype(np.uint32)}
    if label_format == "h5":
        from unpaired_image_captioning_amd.misc.dataloader.label_store import write_hdf5
        path = os.path.join(root, "talk_label.h5")
        write_hdf5(path, arrays)
    else:
        path = os.path.join(root, "talk_label.npz")
        np.savez(path, **arrays)
    return path


def loader_opt(root, label_path, batch_size, seq_per_img, fc_feat_size, att_feat_size, use_box, norm_att_feat, norm_box_feat):
    return argparse.Namespace(batch_size=batch_size, seq_per_img=seq_per_img, fc_feat_size=fc_feat_size,
                              att_feat_size=att_feat_size, use_att=True, use_box=use_box, use_box_cls_prob=0,
                              norm_att_feat=norm_att_feat, norm_box_feat=norm_box_feat, nmt_train_flag=0, nmt_eval_flag=0,
                              input_json=os.path.join(root, "talk.json"), input_label_h5=label_path,
                              input_fc_dir=os.path.join(root, "fc"), input_att_dir=os.path.join(root, "att"),
                              input_box_dir=os.path.join(root, "box"), train_only=0)
