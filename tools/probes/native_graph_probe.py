"""Which HIP graph calls work on a torch-captured hipGraph_t (torch.cuda.CUDAGraph(keep_graph=True).raw_cuda_graph()) on this ROCm:
hipGraphGetNodes, hipGraphClone, hipGraphInstantiate / WithFlags(UseNodePriority), hipGraphKernelNodeSetAttribute(priority),
hipGraphLaunch -- return codes and whether the replay computes the right values (bring-up of csrc/m4d_graph.hip, round 6)."""
import ctypes, torch
hip = ctypes.CDLL("libamdhip64.so")
dev = torch.device("cuda:0")
x = torch.arange(1024, dtype=torch.float32, device=dev)
y = torch.zeros_like(x)
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    y.copy_(x * 2.0 + 1.0)
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph(keep_graph=True)
with torch.cuda.graph(g, stream=s):
    t = x * 2.0
    y.copy_(t + 1.0)
raw = ctypes.c_void_p(int(g.raw_cuda_graph()))
print("raw graph", hex(raw.value))
n = ctypes.c_size_t(0)
print("GetNodes(count)", hip.hipGraphGetNodes(raw, None, ctypes.byref(n)), n.value)
nodes = (ctypes.c_void_p * max(n.value, 1))()
print("GetNodes(list)", hip.hipGraphGetNodes(raw, nodes, ctypes.byref(n)), n.value)
for i in range(n.value):
    ty = ctypes.c_int(-1)
    rc = hip.hipGraphNodeGetType(ctypes.c_void_p(nodes[i]), ctypes.byref(ty))
    print("  node", i, "type rc", rc, "type", ty.value)


def try_exec(label, graph, flags, set_prio=None):
    if set_prio is not None:
        val = (ctypes.c_char * 64)()
        ctypes.cast(val, ctypes.POINTER(ctypes.c_int))[0] = set_prio
        for i in range(n.value):
            ty = ctypes.c_int(-1)
            hip.hipGraphNodeGetType(ctypes.c_void_p(nodes[i]), ctypes.byref(ty))
            if ty.value == 0:
                print(f"  [{label}] SetAttribute(priority={set_prio}) node {i}:", hip.hipGraphKernelNodeSetAttribute(ctypes.c_void_p(nodes[i]), 8, val))
    ex = ctypes.c_void_p()
    hip.hipGraphInstantiateWithFlags.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_void_p, ctypes.c_ulonglong]
    rc = hip.hipGraphInstantiateWithFlags(ctypes.byref(ex), graph, flags)
    print(f"[{label}] InstantiateWithFlags({flags}) rc", rc)
    if rc != 0:
        return
    x.copy_(torch.arange(1024, dtype=torch.float32, device=dev) * 3)
    y.zero_()
    torch.cuda.synchronize()
    rc = hip.hipGraphLaunch(ex, ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
    torch.cuda.synchronize()
    ok = bool(torch.equal(y, x * 2.0 + 1.0))
    print(f"[{label}] Launch rc", rc, "values right:", ok)
    hip.hipGraphExecDestroy(ex)


try_exec("captured graph, plain", raw, 0)
try_exec("captured graph, UseNodePriority flag only", raw, 8)
clone = ctypes.c_void_p()
print("Clone rc", hip.hipGraphClone(ctypes.byref(clone), raw))
try_exec("clone, plain", clone, 0)
lo, hi = ctypes.c_int(0), ctypes.c_int(0)
print("hipDeviceGetStreamPriorityRange", hip.hipDeviceGetStreamPriorityRange(ctypes.byref(lo), ctypes.byref(hi)), "least", lo.value, "greatest", hi.value)
val = (ctypes.c_char * 64)()
print("GetAttribute(priority) node 0:", hip.hipGraphKernelNodeGetAttribute(ctypes.c_void_p(nodes[0]), 8, val), "value", ctypes.cast(val, ctypes.POINTER(ctypes.c_int))[0])
hip.hipGetLastError()
for pr in (-2, -1, 0, 1, 2):
    ctypes.cast(val, ctypes.POINTER(ctypes.c_int))[0] = pr
    rc = hip.hipGraphKernelNodeSetAttribute(ctypes.c_void_p(nodes[0]), 8, val)
    hip.hipGetLastError()                                    # (clear the sticky error: the next framework call would raise it)
    print(f"SetAttribute(hipKernelNodeAttributePriority = {pr}) on a kernel node: rc {rc}")
for attr in (1, 2, 9, 10):
    rc = hip.hipGraphKernelNodeGetAttribute(ctypes.c_void_p(nodes[0]), attr, val)
    hip.hipGetLastError()
    print(f"GetAttribute(id {attr}) rc {rc}")
g.replay(); torch.cuda.synchronize(); print("torch replay of the kept graph right:", bool(torch.equal(y, x * 2.0 + 1.0)))

# ---- do kernel nodes captured on a PRIORITY stream carry that priority?
sp = torch.cuda.Stream(priority=-1)
sp.wait_stream(torch.cuda.current_stream())
g2 = torch.cuda.CUDAGraph(keep_graph=True)
with torch.cuda.graph(g2, stream=sp):
    y.copy_(x * 2.0 + 1.0)
raw2 = ctypes.c_void_p(int(g2.raw_cuda_graph()))
n2 = ctypes.c_size_t(0)
hip.hipGraphGetNodes(raw2, None, ctypes.byref(n2))
nodes2 = (ctypes.c_void_p * max(n2.value, 1))()
hip.hipGraphGetNodes(raw2, nodes2, ctypes.byref(n2))
for i in range(n2.value):
    ty = ctypes.c_int(-1)
    hip.hipGraphNodeGetType(ctypes.c_void_p(nodes2[i]), ctypes.byref(ty))
    if ty.value == 0:
        rc = hip.hipGraphKernelNodeGetAttribute(ctypes.c_void_p(nodes2[i]), 8, val)
        print(f"captured on a priority -1 stream: kernel node {i} GetAttribute(priority) rc {rc} value {ctypes.cast(val, ctypes.POINTER(ctypes.c_int))[0]}")
