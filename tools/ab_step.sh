cp radargnn_amd/librgnn.so /tmp/new.so
one() { python bench.py --no-c4 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1', round(d['ms_per_step'],4), round(d['roofline']['frac'],4))"; }
for i in 1 2 3; do
cp /tmp/new.so radargnn_amd/librgnn.so; one new
cp tools/var/base/librgnn.so radargnn_amd/librgnn.so; one base
done
cp /tmp/new.so radargnn_amd/librgnn.so
