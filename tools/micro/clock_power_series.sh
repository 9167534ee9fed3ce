#!/bin/bash
# Run ON the GPU box (through gpurun): the GPU's shader clock and average power (sysfs hwmon of the device the bench runs on, found by PCI address) every 0.5 s while bench.py runs four workloads.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
bus=$(python3 -c "
import torch
p = torch.cuda.get_device_properties(0)
print('%04x:%02x:%02x.0' % (p.pci_domain_id, p.pci_bus_id, p.pci_device_id))" 2>/dev/null | tail -1)
d=$(for x in /sys/class/drm/card*/device; do grep -q "PCI_SLOT_NAME=$bus" $x/uevent 2>/dev/null && echo $x; done | head -1)
h=$(ls -d $d/hwmon/hwmon* | head -1)
series() { # $1 = seconds
  for i in $(seq 1 $(( $1 * 2 ))); do printf "%s/%s " $(( $(cat $h/freq1_input)/1000000 )) $(( $(cat $h/power1_average 2>/dev/null || cat $h/power1_input)/1000000 )); sleep 0.5; done; echo; }
for args in "--steps 40000" "--sync --steps 20000" "--workload cfg3 --steps 8000" "--workload cfg2 --steps 20000"; do
  echo "== bench.py $args : sclk MHz / power W every 0.5 s"
  (timeout 170 python3 bench.py $args --warmup 5 --no-cpu-baseline --no-also > /tmp/o.json 2>/dev/null) &
  series 30
  wait
  python3 -c "
import json; d=json.loads(open('/tmp/o.json').read().strip().splitlines()[-1]); print('  ->', d['value'], d['ms_per_step'])"
done
