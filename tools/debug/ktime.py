import csv,glob,sys
import numpy as np
pat=sys.argv[2]
f=glob.glob(sys.argv[1]+"/**/*kernel_trace.csv",recursive=True)[0]
rows=list(csv.DictReader(open(f)))
ms=np.array([(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3 for r in rows if pat in r["Kernel_Name"]])
print(len(ms), 'launches of', pat)
step=int(sys.argv[3]) if len(sys.argv)>3 else len(ms)
for i in range(0,len(ms),step): print('  median %.2f us  min %.2f us' % (np.median(ms[i:i+step]), ms[i:i+step].min()))
