import time, torch as pt, os
pt.cuda.init(); pt.zeros(1, device="cuda")
for name, fn in (("current_stream", lambda: pt.cuda.current_stream().cuda_stream), ("raw", lambda: pt._C._cuda_getCurrentRawStream(pt._C._cuda_getDevice())), ("getenv", lambda: os.getenv("PYTORCH_NVML_BASED_CUDA_CHECK")), ("is_available", pt.cuda.is_available)):
    t0 = time.perf_counter()
    for _ in range(1000): fn()
    print(name, (time.perf_counter() - t0) * 1e3, "us per call")
print(len(os.environ))
