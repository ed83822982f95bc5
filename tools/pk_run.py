"""Run a tool against the investigation library tools/pk/libomnifusion_pk.so (tools/pk_build.sh): python tools/pk_run.py tools/conc3.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import omnifusion_amd._lib as L
L.LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "pk", "libomnifusion_pk.so")
sys.argv = sys.argv[1:]
exec(compile(open(sys.argv[0]).read(), sys.argv[0], "exec"))
