"""Driver for the sequential models -- drop-in for test/testSeqRec.py (Caser / AttRec with seq_len 5,
target_len 3; PRME with 1 / 1; DataSet train + test sharing the id maps; SURVEY D3/D7: given flags reach the
model, the two data files are flags, --synthetic writes a time-ordered ml100k-shaped train / test pair)."""
import argparse
import os
import tempfile

from deeprec_amd.models.seq_rec.AttRec import AttRec
from deeprec_amd.models.seq_rec.Caser import Caser
from deeprec_amd.models.seq_rec.PRME import PRME
from deeprec_amd.utils.load_data.load_data_seq import DataSet
from deeprec_amd.utils.load_data.synthetic import write_interaction_log

MODELS = {"Caser": (Caser, 5, 3), "AttRec": (AttRec, 5, 3), "PRME": (PRME, 1, 1)}


def parse_args(argv=None):
    parser = argparse.ArgumentParser(description="DeepRec (MI355X hot path)")
    parser.add_argument("--model", choices=sorted(MODELS), default="AttRec")
    parser.add_argument("--epochs", type=int, default=None)
    parser.add_argument("--num_factors", type=int, default=None)
    parser.add_argument("--display_step", type=int, default=None)
    parser.add_argument("--batch_size", type=int, default=None)
    parser.add_argument("--learning_rate", type=float, default=None)
    parser.add_argument("--reg_rate", type=float, default=None)
    parser.add_argument("--train", default="../data/ml100k/temp/train.dat")
    parser.add_argument("--test", default="../data/ml100k/temp/test.dat")
    parser.add_argument("--synthetic", action="store_true")
    parser.add_argument("--device", default=None)
    return parser.parse_args(argv)


def _split_by_time(src, train_path, test_path, frac=0.8):
    """Each user's first `frac` interactions (by time) -> train, the rest -> test; test lines whose item never
    occurs in train are dropped (the model has no row for them)."""
    per = {}
    with open(src) as f:
        for line in f:
            u, i, r, t = line.rstrip("\n").split("\t")
            per.setdefault(u, []).append((int(t), i, line))
    train_items, held = set(), []
    with open(train_path, "w") as ftr:
        for u, rows in per.items():
            rows.sort()
            cut = max(1, int(len(rows) * frac))
            for x, (_, i, line) in enumerate(rows):
                if x < cut:
                    ftr.write(line)
                    train_items.add(i)
                else:
                    held.append((i, line))
    with open(test_path, "w") as fte:
        for i, line in held:
            if i in train_items:
                fte.write(line)


def main(argv=None):
    args = parse_args(argv)
    cls, seq_len, target_len = MODELS[args.model]
    header = ["user", "item", "rating", "time"]
    with tempfile.TemporaryDirectory() as tmp:
        tr, te = args.train, args.test
        if args.synthetic:
            write_interaction_log(os.path.join(tmp, "all.dat"))
            tr, te = os.path.join(tmp, "train.dat"), os.path.join(tmp, "test.dat")
            _split_by_time(os.path.join(tmp, "all.dat"), tr, te)
        train_data = DataSet(path=tr, sep="\t", header=header, isTrain=True, seq_len=seq_len, target_len=target_len)
        # num_users / num_items: the test matrix gets the train shape (train.num_item already counts the pad item 0;
        # without them the reference sizes it by len(item_map), one short of the largest shifted id)
        test_data = DataSet(path=te, sep="\t", header=header, user_map=train_data.user_map, item_map=train_data.item_map,
                            num_users=train_data.num_user, num_items=train_data.num_item)
    kw = {k: v for k, v in (("learning_rate", args.learning_rate), ("reg_rate", args.reg_rate), ("epoch", args.epochs),
                            ("batch_size", args.batch_size), ("display_step", args.display_step),
                            ("device", args.device)) if v is not None}
    model = cls(train_data.num_user, train_data.num_item, **kw)
    model.build_network(L=train_data.sequences.L, num_T=train_data.sequences.T,
                        **({"num_factor": args.num_factors} if args.num_factors else {}))
    model.execute(train_data, test_data)
    return model


if __name__ == "__main__":
    main()
