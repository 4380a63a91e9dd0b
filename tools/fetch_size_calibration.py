"""FETCH_SIZE per launch of tools/micro/fetch_size_calibration.hip's kernels against the bytes each asked for (see that file).

    python tools/fetch_size_calibration.py <dir of the rocprofv3 --pmc FETCH_SIZE pass>
"""
import collections
import csv
import glob
import sys

path = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)[0]
values = collections.defaultdict(list)
for row in csv.DictReader(open(path)):
    if row["Counter_Name"] == "FETCH_SIZE":
        values[row["Kernel_Name"].split("(")[0]].append(float(row["Counter_Value"]))
N = 1 << 20
asked = {"stream_kernel<HIP_vector_type<unsigned int, 4": 16 * N, "stream_kernel<HIP_vector_type<unsigned int, 2": 8 * N, "stream_kernel<unsigned int>": 4 * N,
         "stream_kernel<unsigned char>": N, "gather_kernel": 4 * N + 65536 * 8, "three_streams_kernel": 13 * N}
for name, launches in values.items():
    want = next((v for k, v in asked.items() if k in name), None)
    if want is None:
        continue
    cold = launches[0::2] if ("gather" in name or "three" in name) else launches
    warm = launches[1::2] if ("gather" in name or "three" in name) else []
    def line(tag, xs):
        if xs:
            mean = sum(xs) / len(xs) * 1024
            print("%-70s %-5s FETCH_SIZE %10.3f MB  asked for %8.3f MB  ratio %.3f  (%d launches)" % (name[:70], tag, mean / 1e6, want / 1e6, mean / want, len(xs)))
    line("cold", cold)
    line("warm", warm)
