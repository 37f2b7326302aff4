import json,sys
a=json.load(open(sys.argv[1])); b=json.load(open(sys.argv[2]))
ta=tb=0
for x,y in zip(a,b):
    if x['tile']!=y['tile']:
        print(f"{x['name']:12s} {x['Cin']:4d}->{x['Cout']:4d} k{x['k']} @{x['H']:3d} {x['tile']:12s} {x['us']:7.1f} -> {y['tile']:12s} {y['us']:7.1f}")
    ta+=x['us']; tb+=y['us']
print('totals',ta,tb)
