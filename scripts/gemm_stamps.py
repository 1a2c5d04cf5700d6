import ctypes as C, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from ekf_vio_amd import TightlyCoupledEKF
g = TightlyCoupledEKF(max_features=4, hooks=True)
for variant in [int(a) + 10000 for a in (sys.argv[1:] or ['1', '48'])]:
    buf = (C.c_double * 41)()
    rc = g.lib.ekfvio_test_gemm_bench(g.h, 1, 0, 790, 790, 512, 20, variant, buf)
    v = [int(x) for x in buf[1:41]]
    t0 = v[0]
    print("variant", variant - 10000, "rc", rc, "mean_us", buf[0])
    print("  prologue            %6d" % (v[1] - v[0]))
    prev = v[1]
    for i in range(2, 36):
        if v[i] > 0:
            print("  tile stamp %2d        %6d" % (i - 2, v[i] - prev)); prev = v[i]
    print("  tail tiles+->loop end %6d" % (v[36] - prev))
    print("  reduce              %6d" % (v[37] - v[36]))
    print("  epilogue            %6d" % (v[38] - v[37]))
    print("  total               %6d cycles in the stamped workgroup; launch-to-launch %.2f us -> the kernel body is %.2f us at 2.4 GHz" % (v[38] - v[0], buf[0], (v[38] - v[0]) / 2400.0))
