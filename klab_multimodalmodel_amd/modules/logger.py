"""Console + `<result_dir>/train.log` logger (ref/modules/logger.py:4-20)."""
import logging
import os


def get_logger(args):
    logger = logging.getLogger("klab_train")
    logger.setLevel(logging.INFO)
    fmt = logging.Formatter('%(asctime)s - %(levelname)s - %(message)s')
    if not logger.handlers:
        for h in (logging.StreamHandler(), logging.FileHandler(os.path.join(args.result_dir, 'train.log'))):
            h.setLevel(logging.INFO)
            h.setFormatter(fmt)
            logger.addHandler(h)
    return logger
