"""NUMA-local ranks, checked WITHOUT hardware on made-up sysfs trees: a batch's host side -- its
parser threads and its pinned staging memory -- belongs on the NUMA node its GPU hangs off; ranks that share a node take
disjoint slices of its cores.  (h263-rs_amd/csrc/worker_pool.cpp: host_placement; the same code a batch runs when it is made.)
The topologies use the CPUs this process may run on, because a placement is always cut down to the affinity mask."""
import json
import os
import subprocess
import sys

import pytest

import h263mi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CPUS = sorted(os.sched_getaffinity(0))


def make_sysfs(root, node_cpus, device_nodes, siblings=None):
    """root/devices/system/node/nodeK/cpulist, root/bus/pci/devices/<bdf>/numa_node, optional thread_siblings_list"""
    ids = []
    for k, cpus in enumerate(node_cpus):
        d = os.path.join(root, "devices", "system", "node", "node%d" % k)
        os.makedirs(d)
        open(os.path.join(d, "cpulist"), "w").write(",".join(str(c) for c in cpus) + "\n")
    for i, node in enumerate(device_nodes):
        bdf = "0000:%02x:00.0" % (0x10 + i)
        d = os.path.join(root, "bus", "pci", "devices", bdf)
        os.makedirs(d)
        open(os.path.join(d, "numa_node"), "w").write("%d\n" % node)
        ids.append(bdf.upper() if i % 2 else bdf)                 # (hipDeviceGetPCIBusId may print hex digits in upper case)
    for cpu, sib in (siblings or {}).items():
        d = os.path.join(root, "devices", "system", "cpu", "cpu%d" % cpu, "topology")
        os.makedirs(d)
        open(os.path.join(d, "thread_siblings_list"), "w").write(sib + "\n")
    return ids


@pytest.mark.skipif(len(CPUS) < 8, reason="needs 8 CPUs in the affinity mask")
def test_eight_ranks_on_two_sockets_get_eight_disjoint_cpu_sets(tmp_path):
    c = CPUS[:8]
    ids = make_sysfs(str(tmp_path), [c[:4], c[4:]], [0, 0, 0, 0, 1, 1, 1, 1])
    sets = []
    for dev in range(8):
        node, cpus = h263mi.debug_host_placement(ids, dev, 8, str(tmp_path))
        assert node == (0 if dev < 4 else 1)
        assert cpus and set(cpus) <= set(c[:4] if dev < 4 else c[4:])
        sets.append(set(cpus))
    assert all(sets[i].isdisjoint(sets[j]) for i in range(8) for j in range(i)), sets
    assert set().union(*sets) == set(c)
    # ONE rank on the same host (a service process, h263mi_set_ranks_per_node(1)): the whole node of its GPU
    node, cpus = h263mi.debug_host_placement(ids, 5, 1, str(tmp_path))
    assert node == 1 and cpus == c[4:]
    # two ranks (devices 0 and 1 hang off node 0 both): halves of node 0
    a, b = (h263mi.debug_host_placement(ids, d, 2, str(tmp_path))[1] for d in (0, 1))
    assert a == c[:2] and b == c[2:4]


@pytest.mark.skipif(len(CPUS) < 8, reason="needs 8 CPUs in the affinity mask")
def test_slices_are_cut_out_of_cores_not_hyperthreads(tmp_path):
    """siblings are numbered far apart (k and k + cores): a rank's slice holds whole cores"""
    c = CPUS[:8]
    sib = {c[k]: "%d,%d" % (c[k % 4], c[k % 4 + 4]) for k in range(8)}          # 4 cores x 2 threads, one node
    ids = make_sysfs(str(tmp_path), [c], [0, 0], siblings=sib)
    a = h263mi.debug_host_placement(ids, 0, 2, str(tmp_path))[1]
    b = h263mi.debug_host_placement(ids, 1, 2, str(tmp_path))[1]
    assert a == sorted([c[0], c[1], c[4], c[5]]) and b == sorted([c[2], c[3], c[6], c[7]])


def test_unknown_topology_and_the_off_switch_place_nothing(tmp_path):
    assert h263mi.debug_host_placement(["0000:99:00.0"], 0, 1, str(tmp_path)) == (-1, [])      # no such device in sysfs
    ids = make_sysfs(str(tmp_path), [CPUS], [-1])                                               # numa_node = -1 (single socket)
    assert h263mi.debug_host_placement(ids, 0, 1, str(tmp_path)) == (-1, [])
    code = ("import sys; sys.path.insert(0, %r); import h263mi; print(h263mi.debug_host_placement(%r, 0, 1, %r))"
            % (os.path.join(ROOT, "h263-rs_amd"), ["0000:10:00.0"], str(tmp_path / "t2")))
    ids2 = make_sysfs(str(tmp_path / "t2"), [CPUS], [0])
    on = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=dict(os.environ)).stdout
    off = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=dict(os.environ, H263MI_NUMA="0")).stdout
    assert on.strip() == repr((0, CPUS)) and off.strip() == repr((-1, []))
    forced = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True,
                            env=dict(os.environ, H263MI_NUMA_NODE="0", H263MI_SYSFS_ROOT=str(tmp_path / "t2"))).stdout
    assert forced.strip() == repr((0, CPUS))


@pytest.mark.skipif(len(CPUS) < 8, reason="needs 8 CPUs in the affinity mask")
def test_eight_gloo_ranks_of_the_bench_stand_in_report_disjoint_cpu_sets(tmp_path):
    """the launcher path of a first 8-GPU run: 8 ranks of bench.py's CPU stand-in (gloo), each asking the LIBRARY where its
    batch would be placed on a made-up 2-socket / 8-GPU host -- 8 disjoint CPU sets, four per socket"""
    c = CPUS[:8]
    make_sysfs(str(tmp_path), [c[:4], c[4:]], [0, 0, 0, 0, 1, 1, 1, 1])
    env = dict(os.environ, H263MI_BENCH_STUB="1", H263MI_SYSFS_ROOT=str(tmp_path), H263MI_STUB_PCI_IDS=",".join(
        "0000:%02x:00.0" % (0x10 + i) for i in range(8)))
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "1", "--warmup", "0"], env=env,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    out = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][0])
    pl = out["placement_per_rank"]
    assert [p["node"] for p in pl] == [0, 0, 0, 0, 1, 1, 1, 1]
    sets = [set(p["cpus"]) for p in pl]
    assert all(s for s in sets) and all(sets[i].isdisjoint(sets[j]) for i in range(8) for j in range(i)), sets
