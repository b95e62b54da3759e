#!/usr/bin/env python3
"""python train.py [use_gpu console_mode show_progress_bar save_train_progress]

Same positional arguments as the reference's train.py (README.md:94-105); the Socket.IO relay of the
web UI is out of scope, so the run is always in console mode.  Trains the nets on seeded synthetic
pages through the HIP backend and maintains model_weights.json."""
import sys


def as_bool(arg):
    return {'true': True, 'false': False}.get(str(arg).lower(), arg)


def main(use_gpu=True, console_mode=True, show_progress_bar=False, save_train_progress=False):
    from univer_ocr_amd.my_model.train import train_model
    print('Running in console mode')
    try:
        train_model(as_bool(use_gpu), as_bool(show_progress_bar), as_bool(save_train_progress))
    except KeyboardInterrupt:
        print('Stopped by keyboard interrupt')


if __name__ == '__main__':
    main(*sys.argv[1:])
