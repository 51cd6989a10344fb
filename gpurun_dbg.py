import os, ctypes as C, numpy as np
os.environ["LVK_LIB"] = os.path.join(os.path.dirname(os.path.abspath(__file__)), "larvio_amd", "liblvk_hip_timing.so")
import larvio_amd
from larvio_amd import synthetic as S
from larvio_amd._lib import lib
from tests.conftest import synth_frames
frames = synth_frames(40, 30)
seq = S.imu_only_sequence()
ctx = larvio_amd.Context(0)
fe = larvio_amd.ImageProcessor(S.frontend_config(max_features_num=150), ctx); fe.initialize()
L = lib()
k_first = max(int(frames[0][0] * 200) - 2, 0)
imu_all = seq.imu_array(k_first, k_first + 800)
tk = np.zeros(32, np.uint64)
for i, (ts, img) in enumerate(frames):
    buf = imu_all[:int(np.searchsorted(imu_all["t"], ts + 0.05))]
    fe.processImage(img, buf[-60:], ts=ts)
    L.lvk_debug_fm_ticks(tk.ctypes.data_as(C.c_void_p))
    if i >= 20:
        t = tk[:13].astype(np.int64)
        print("mode", tk[18], "m", tk[16], "iters", tk[17], "compact %.2f undist %.2f | subsets %.2f solve %.2f (hh %.2f q %.2f fin %.2f) score %.2f replay %.2f | mask %.2f commit %.2f total %.2f us" % (
            (t[1]-t[0])/100, (t[2]-t[1])/100, (t[5]-t[4])/100, (t[6]-t[5])/100, (t[11]-t[5])/100, (t[12]-t[11])/100, (t[6]-t[12])/100, (t[7]-t[6])/100, (t[8]-t[7])/100, (t[9]-t[8])/100, (t[10]-t[9])/100, (t[10]-t[0])/100))
