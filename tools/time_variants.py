import sys, os, shutil, subprocess
# time the bench workload with alternative builds of the library (experiment only)
for name in sys.argv[1:]:
    print("==", name, flush=True)
    env = dict(os.environ, RT_LIB_OVERRIDE=os.path.abspath(name))
    subprocess.run([sys.executable, "-u", "tools/time_one.py"], env=env)
