"""Driver for the rating-prediction models -- drop-in for test/test_rating_pred.py (MF / NNMF on
load_data_rating, NFM on load_data_fm; SURVEY D3/D7: given flags reach the model, data paths are flags,
--synthetic generates ml100k- / frappe-shaped files)."""
import argparse
import os
import tempfile

from deeprec_amd.models.rating_prediction.mf import MF
from deeprec_amd.models.rating_prediction.nfm import NFM
from deeprec_amd.models.rating_prediction.nnmf import NNMF
from deeprec_amd.utils.load_data.load_data_content import load_data_fm
from deeprec_amd.utils.load_data.load_data_rating import load_data_rating
from deeprec_amd.utils.load_data.synthetic import write_interaction_log, write_libfm


def parse_args(argv=None):
    parser = argparse.ArgumentParser(description="nnRec (MI355X hot path)")
    parser.add_argument("--model", choices=["MF", "NNMF", "NFM"], default="MF")
    parser.add_argument("--epochs", type=int, default=None)
    parser.add_argument("--batch_size", type=int, default=None)
    parser.add_argument("--learning_rate", type=float, default=None)
    parser.add_argument("--reg_rate", type=float, default=None)
    parser.add_argument("--num_factors", type=int, default=None)
    parser.add_argument("--display_step", type=int, default=None)
    parser.add_argument("--show_time", action="store_true")
    parser.add_argument("--T", type=int, default=None)
    parser.add_argument("--data", default="../data/ml100k/movielens_100k.dat")
    parser.add_argument("--fm_dir", default="../data/frappe/", help="directory of frappe.{train,test}.libfm")
    parser.add_argument("--synthetic", action="store_true")
    parser.add_argument("--device", default=None)
    return parser.parse_args(argv)


def main(argv=None):
    args = parse_args(argv)
    kw = {k: v for k, v in (("learning_rate", args.learning_rate), ("reg_rate", args.reg_rate), ("epoch", args.epochs),
                            ("batch_size", args.batch_size), ("display_step", args.display_step), ("T", args.T),
                            ("device", args.device)) if v is not None}
    if args.show_time:
        kw["show_time"] = True
    with tempfile.TemporaryDirectory() as tmp:
        if args.model == "NFM":
            fm_dir = args.fm_dir
            if args.synthetic:
                write_libfm(os.path.join(tmp, "frappe.train.libfm"), 20000, seed=0)
                write_libfm(os.path.join(tmp, "frappe.test.libfm"), 2000, seed=1)
                fm_dir = tmp + os.sep
            train_data, test_data, feature_M = load_data_fm(path=fm_dir)
            model = NFM(957, 4082, **kw)                        # test_rating_pred.py: n_user / n_item are nominal here
            model.build_network(feature_M, **({"num_factor": args.num_factors} if args.num_factors else {}))
        else:
            path = args.data
            if args.synthetic:
                path = os.path.join(tmp, "log.dat")
                write_interaction_log(path)
            train_data, test_data, n_user, n_item = load_data_rating(path=path, header=["user_id", "item_id", "rating", "t"],
                                                                     test_size=0.1, sep="\t")
            model = {"MF": MF, "NNMF": NNMF}[args.model](n_user, n_item, **kw)
            if args.model == "MF":
                model.build_network(**({"num_factor": args.num_factors} if args.num_factors else {}))
            else:
                model.build_network()
    model.execute(train_data, test_data)
    return model


if __name__ == "__main__":
    main()
