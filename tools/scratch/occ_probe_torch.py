# the occupancy probe under the HIP runtime torch bundles (the runtime the Python benches use), next to the stand-alone binary
import ctypes, os, sys
import torch
torch.zeros(1, device="cuda")
lib = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "libocc_probe.so"))
sys.stdout.flush()
lib.occ_probe_main()
