"""Drop-in for the reference's ``model`` package: put this directory FIRST on PYTHONPATH and the
reference drivers' ``from model import nets`` resolves to the MI355X implementation."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from cellsegmentation_amd.model import *  # noqa: F401,F403,E402
from cellsegmentation_amd.model import nets  # noqa: F401,E402
