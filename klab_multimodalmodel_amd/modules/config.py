"""Command-line surface of the reference's trainer (ref/modules/config.py:3-24): same 15 flags,
same defaults, same choices -- so `ref/train.py` parses identically on top of this package."""
import argparse

T5_NAMES = ['t5-small', 't5-base', 't5-large', 't5-3b', 't5-11b']

# (flag, kwargs) in the reference's order
_FLAGS = [
    ('--image_model_name', dict(type=str, default="microsoft/swinv2-base-patch4-window8-256")),
    ('--image_model_train', dict(action='store_true')),
    ('--language_model_name', dict(type=str, default='t5-large', choices=T5_NAMES)),
    ('--transformer_model_name', dict(type=str, default='t5-large', choices=T5_NAMES)),
    ('--max_source_length', dict(type=int, default=256)),
    ('--max_target_length', dict(type=int, default=128)),
    ('--lr', dict(type=float, default=0.001)),
    ('--lr_scheduler', dict(type=str, default='', choices=['', 'cosine', 'linear', 'exponential', 'step'])),
    ('--batch_size', dict(type=int, default=64)),
    ('--accumulation_steps', dict(type=int, default=1)),
    ('--num_epochs', dict(type=int, default=None)),
    ('--num_steps', dict(type=int, default=None)),
    ('--save_interval', dict(type=int, default=None)),
    ('--data_dir', dict(type=str, default='/user/data/mscoco2017/')),
    ('--result_dir', dict(type=str, default='results/')),
]


def build_parser(allow_local_dirs=False):
    """allow_local_dirs drops the hub-name `choices` so that local checkpoint directories can be passed
    (there is no hub access offline); everything else is the reference's parser."""
    parser = argparse.ArgumentParser(description='Swin-V2 -> T5 caption training (MI355X-native engine)')
    for flag, kw in _FLAGS:
        kw = dict(kw)
        if allow_local_dirs:
            kw.pop('choices', None) if flag in ('--language_model_name', '--transformer_model_name') else None
        parser.add_argument(flag, **kw)
    return parser


def parse_arguments(argv=None):
    import os
    return build_parser(allow_local_dirs=os.environ.get("KLAB_LOCAL_MODEL_DIRS", "0") == "1").parse_args(argv)
