"""Command line entry, same flags as the reference's run_knowledgable_recommendation.py (single-dash gflags style)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

from jTransUP.models import knowledgable_recommendation
from jTransUP.models.base import flag_defaults, get_flags
from jTransUP.utils.flags import FLAGS

if __name__ == '__main__':
    get_flags()
    FLAGS(sys.argv)
    flag_defaults(FLAGS)
    knowledgable_recommendation.run(only_forward=FLAGS.eval_only_mode)
