"""Shared launcher of the three run_*.py command lines."""
import importlib
import sys


def main(task):
    """task: 'item_recommendation' | 'knowledge_representation' | 'knowledgable_recommendation'."""
    from jTransUP.models.base import flag_defaults, get_flags
    from jTransUP.utils.flags import FLAGS
    driver = importlib.import_module('jTransUP.models.' + task)
    get_flags()
    FLAGS(sys.argv)
    flag_defaults(FLAGS)
    driver.run(only_forward=FLAGS.eval_only_mode)
