"""Command line of the reference's ``main.py`` (flags at main.py:1463-1533, dispatch at :1535-1570):

    python main.py -train --model DIR
    python main.py -process [-concatenated] [--binarymaskthreshold t] --model DIR --source-imgs S --mask-output-imgs R

Every flag of the reference parses (same names, defaults and the ``type=bool`` quirk: ``-cload False`` is
still True, as in the reference); flags whose code path is outside this build raise NotImplementedError
when reached instead of being silently ignored."""
import argparse


def build_parser():
    p = argparse.ArgumentParser()
    for flag in ("-train", "-cleaned", "-frozen", "-clippify", "-debug", "-noinject", "-freeze", "-viscritic",
                 "-vismasker", "-visdataset", "-trunk", "-higheval", "-separate", "-salience", "-process_salience",
                 "-grabcut", "-crf", "-directeval", "-soft", "-resimages", "-noevalmode", "-eval", "-process", "-test",
                 "-concatenated", "-softmask"):
        p.add_argument(flag, action="store_true")
    # (this build's own switch, not a flag of the reference) -process / -eval with fp16 activations and weights, fp32 accumulation
    p.add_argument("-fp16", action="store_true")
    for flag in ("-masker", "-critic", "-cload", "-mload", "-staticnorm", "-visbesteval", "-salglobal"):
        p.add_argument(flag, type=bool, default=True)
    p.add_argument("--salience-thresh", type=float, default="1.5")
    p.add_argument("--eval-thresh", type=float, default=0.05)
    p.add_argument("--dropout", type=float, default=0.3)
    p.add_argument("--lr", type=float, default=0.00005)   # parsed, never read (as in the reference)
    p.add_argument("--threshrew", type=float, default=0)
    p.add_argument("--trainasvis", type=int, default=0)
    p.add_argument("--false", type=bool, default=False)
    p.add_argument("--envname", type=str, default="Treechop")
    p.add_argument("--visname", type=str, default="curves")
    p.add_argument("--datamode", type=str, default="trunk")
    p.add_argument("--purevis", type=str, default="")
    p.add_argument("--sortidx", type=int, default=1)
    p.add_argument("--chfak", type=int, default=1)
    p.add_argument("--shift", type=int, default=12)
    p.add_argument("--lfak", type=int, default=5)
    p.add_argument("--neck", type=int, default=32)
    p.add_argument("--clossfak", type=int, default=5)
    p.add_argument("--cepochs", type=int, default=15)
    p.add_argument("--mepochs", type=int, default=1)
    p.add_argument("--high-rew-thresh", type=float, default=0.7)
    p.add_argument("--low-rew-thresh", type=float, default=0.3)
    p.add_argument("--L2", type=float, default=0.0)
    p.add_argument("--L1", type=float, default=0.5)
    p.add_argument("--saveevery", type=int, default=5)
    p.add_argument("--visevery", type=int, default=100)
    p.add_argument("--rewidx", type=int, default=1)
    p.add_argument("--gammas", type=str, default="0.98-0.97-0.96-0.95")
    p.add_argument("--testsize", type=int, default=5000)
    p.add_argument("--datasize", type=int, default=100000)
    p.add_argument("--name", type=str, default="default-model")
    p.add_argument("--model", type=str, default="default-model")
    p.add_argument("--runs", type=int, default=1)
    p.add_argument("--source-imgs", type=str, default="")
    p.add_argument("--mask-output-imgs", type=str, default="results")
    p.add_argument("--output-video", type=str, default="")
    p.add_argument("--binarymaskthreshold", type=float, default=0.5)
    return p


def parse_args(argv=None):
    args = build_parser().parse_args(argv)
    args.workers = (1, 1, 1)
    args.live = not args.frozen
    args.inject = not args.noinject
    args.name = args.model
    if args.test:
        args.eval = True
        args.train = True if not args.cload else False
        args.visbesteval = True
        args.crf = False
        args.salience = True
    return args


def main(argv=None):
    from .handler import Handler
    args = parse_args(argv)
    H = Handler(args)
    if args.train:
        H.load_data()
    if args.trainasvis:
        raise NotImplementedError("--trainasvis (dataset visualisation) is outside this build's scope")
    if args.cload:
        H.load_models(modelnames=[H.criticname])
    if args.mload:
        H.load_models(modelnames=[H.maskername])
    if args.train:
        if args.critic:
            H.critic_pipe(mode="train")
            H.save_models(modelnames=[H.criticname])
        if args.masker:
            H.segmentation_training()
            H.save_models(modelnames=[H.maskername])
    if args.eval:
        H.eval()
    if args.viscritic or args.vismasker:
        raise NotImplementedError("-viscritic / -vismasker (videos) are outside this build's scope")
    if args.process:
        H.segment(folder=args.source_imgs)
    return H
