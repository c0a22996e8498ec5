"""Driver for the item-ranking models -- drop-in for test/test_item_ranking.py (same flags, same flow:
load_data_neg -> Model(n_user, n_item) -> build_network() -> execute(train, test)).

Differences, on purpose: a flag that is given is passed to the model (the reference parses its
flags and then ignores them; a flag that is NOT given leaves the model's own default in force, which is what
the reference's behaviour amounts to); the data file is a flag, and --synthetic writes an ml100k-shaped log
first because the reference's data directory does not travel with this repository; the model list is the
§8 scope.
"""
import argparse
import os
import tempfile

from deeprec_amd.models.item_ranking.bprmf import BPRMF
from deeprec_amd.models.item_ranking.cml import CML
from deeprec_amd.models.item_ranking.gmf import GMF
from deeprec_amd.models.item_ranking.neumf import NeuMF
from deeprec_amd.utils.load_data.load_data_ranking import load_data_neg
from deeprec_amd.utils.load_data.synthetic import write_interaction_log

MODELS = {"CML": CML, "NeuMF": NeuMF, "GMF": GMF, "BPRMF": BPRMF}


def parse_args(argv=None):
    parser = argparse.ArgumentParser(description="DeepRec (MI355X hot path)")
    parser.add_argument("--model", choices=sorted(MODELS), default="BPRMF")
    parser.add_argument("--epochs", type=int, default=None)
    parser.add_argument("--num_factors", type=int, default=None)
    parser.add_argument("--display_step", type=int, default=None)
    parser.add_argument("--batch_size", type=int, default=None)
    parser.add_argument("--learning_rate", type=float, default=None)
    parser.add_argument("--reg_rate", type=float, default=None)
    parser.add_argument("--data", default="../data/ml100k/movielens_100k.dat")
    parser.add_argument("--synthetic", action="store_true", help="generate an ml100k-shaped log instead of --data")
    parser.add_argument("--device", default=None)
    return parser.parse_args(argv)


def main(argv=None):
    args = parse_args(argv)
    path = args.data
    tmp = None
    if args.synthetic:
        tmp = tempfile.NamedTemporaryFile("w", suffix=".dat", delete=False)
        tmp.close()
        write_interaction_log(tmp.name)
        path = tmp.name
    try:
        train_data, test_data, n_user, n_item = load_data_neg(path=path, test_size=0.2, sep="\t")
    finally:
        if tmp is not None:
            os.unlink(tmp.name)
    kw = {k: v for k, v in (("learning_rate", args.learning_rate), ("reg_rate", args.reg_rate), ("epoch", args.epochs),
                            ("batch_size", args.batch_size), ("display_step", args.display_step),
                            ("device", args.device)) if v is not None}
    model = MODELS[args.model](n_user, n_item, **kw)
    model.build_network(**({"num_factor": args.num_factors} if args.num_factors is not None else {}))
    model.execute(train_data, test_data)
    return model


if __name__ == "__main__":
    main()
