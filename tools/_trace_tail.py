import csv, json, sys
f, out = sys.argv[1], sys.argv[2]
rows = [r for r in csv.DictReader(open(f))]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = rows[-3200:]
json.dump([(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:100], r.get("Stream_Id", ""), r.get("Queue_Id", "")) for r in rows], open(out, "w"))
print(len(rows))
