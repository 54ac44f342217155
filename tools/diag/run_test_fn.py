import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]
import importlib, traceback
import deep_q_learning_amd as dq
if os.environ.get("CNN_FLAGS"):
    _init = dq.CnnEngine.__init__
    def init(self, *a, **k):
        _init(self, *a, **k); self.set_flags(int(os.environ["CNN_FLAGS"]))
    dq.CnnEngine.__init__ = init
mod = importlib.import_module(sys.argv[1])
try:
    getattr(mod, sys.argv[2])(dq, *[eval(a) for a in sys.argv[3:]])
    print("PASSED")
except AssertionError as e:
    traceback.print_exc(limit=2)
