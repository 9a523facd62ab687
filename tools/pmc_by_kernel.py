"""Sum a rocprofv3 --pmc counter per kernel name.  usage: pmc_by_kernel.py <counter_collection.csv> <out.csv>"""
import sys
import pandas as pd

d = pd.read_csv(sys.argv[1])
name = "Kernel_Name" if "Kernel_Name" in d.columns else "Kernel Name"
val = "Counter_Value" if "Counter_Value" in d.columns else "Counter Value"
d[name] = d[name].str.slice(0, 48)
g = d.groupby(name).agg(Counter_Value=(val, "sum"), Dispatches=(val, "size")).reset_index()
g.to_csv(sys.argv[2], index=False)
print(g.sort_values("Counter_Value", ascending=False).head(12).to_string())
