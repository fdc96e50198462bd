import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "scripts"))
sys.argv = ["x"]
import importlib.util
spec = importlib.util.spec_from_file_location("mb", os.path.join(ROOT, "scripts", "microbench.py"))
mb = importlib.util.module_from_spec(spec); spec.loader.exec_module(mb)
print(os.environ.get("CALM_VIT_LIB", "default lib"))
for B, S in ((256, 224), (256, 176), (256, 128), (256, 80)):
    mb.cnn(B, S)
