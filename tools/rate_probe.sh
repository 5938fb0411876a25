set -e
cd $GRAFT_REPO_ROOT
H=cal_22-mpc_amd/host
SRCS=$(ls $H/*.cpp | grep -v main.cpp)
hipcc -O2 -std=c++17 -I $H -I include -o /tmp/perline_probe tests/native/perline_probe.cpp $SRCS -L cal_22-mpc_amd -lmpc_hip -Wl,-rpath,$PWD/cal_22-mpc_amd
python - <<'PY'
import importlib,sys
sys.path.insert(0,'.')
T=importlib.import_module("cal_22-mpc_amd.traces"); C=importlib.import_module("cal_22-mpc_amd.configs")
T.save_npy("/tmp/t.npy", T.mixed(2000000,64)); C.write_config(C.probe_config(64), "/tmp/c.json")
PY
for buf in 0 4096 65536 1048576; do
  if [ $buf = 0 ]; then A=""; else A=$buf; fi
  /tmp/perline_probe VPC /tmp/c.json /tmp/t.npy /tmp/r.csv /tmp/d.csv /tmp/s.bin $A
  /tmp/perline_probe BDI - /tmp/t.npy /tmp/r2.csv /tmp/d2.csv /tmp/s2.bin $A
done
