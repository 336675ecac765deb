"""python run_knowledgable_recommendation.py -model_type ... : the reference's command line on the MI355X package."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

if __name__ == '__main__':
    from jTransUP.cli import main
    main('knowledgable_recommendation')
