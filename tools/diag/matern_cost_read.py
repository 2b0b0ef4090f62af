import csv, sys
US = [0.05, 0.5, 1.0, 1.9, 2.1, 3.0, 4.0, 6.0, 10.0, 19.0, 21.0, 40.0, 80.0]
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "matern_points" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
N = 1 << 22
for u, r in zip(US, rows):
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    print("u = %6.2f : %8.1f us  -> %.2f G points/s" % (u, d, N / d / 1e3))
