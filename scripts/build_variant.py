import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mltcnn_pkg, os, sys
pkg = mltcnn_pkg.load()
d = os.path.join(os.path.dirname(pkg.build.__file__), "_variants")
name = sys.argv[1]
pkg.build.build_lib(force=True, defines=sys.argv[2:], out=os.path.join(d, f"lib_{name}.so"))
print(name)
