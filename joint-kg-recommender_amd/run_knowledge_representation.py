"""Command line entry, same flags as the reference's run_knowledge_representation.py (single-dash gflags style)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

from jTransUP.models import knowledge_representation
from jTransUP.models.base import flag_defaults, get_flags
from jTransUP.utils.flags import FLAGS

if __name__ == '__main__':
    get_flags()
    FLAGS(sys.argv)
    flag_defaults(FLAGS)
    knowledge_representation.run(only_forward=FLAGS.eval_only_mode)
