"""Package power and shader clock of the GPUs a bench run uses, sampled once a second beside the timed launches — by RANK 0 ONLY, for every GPU
of the job in one pass. Source: the amdgpu hwmon files in sysfs (no child process inside the timed region); `rocm-smi` (one call for all devices)
only where sysfs is not readable. The C2 throughput launch runs the package AT ITS POWER CAP (profiles/r3_power.txt): the figure that explains the
clock the kernel is held at; with N GPUs the spread over them shows whether the node holds them lower than a lone GPU. Any failure leaves nulls."""
import glob
import os
import re
import subprocess
import threading
import time

import numpy as np


def _read(path):
    try:
        with open(path) as f:
            return f.read().strip()
    except Exception:
        return None


def amdgpu_cards():
    """[(pci address, hwmon dir, device dir)] of the amdgpu cards visible in sysfs, ordered by PCI address"""
    out = []
    for dev in glob.glob("/sys/class/drm/card[0-9]*/device"):
        if _read(os.path.join(dev, "vendor")) != "0x1002":
            continue
        hw = glob.glob(os.path.join(dev, "hwmon", "hwmon*"))
        if not hw:
            continue
        out.append((os.path.basename(os.path.realpath(dev)), hw[0], dev))
    return sorted(set(out))


def _card_sample(hw, dev):
    """(sclk MHz, package W) of one card from sysfs, or None"""
    w = None
    for name in ("power1_average", "power1_input"):
        v = _read(os.path.join(hw, name))
        if v and v.isdigit():
            w = int(v) * 1e-6
            break
    mhz = None
    v = _read(os.path.join(hw, "freq1_input"))
    if v and v.isdigit():
        mhz = int(v) * 1e-6
    else:
        m = re.search(r"(\d+)Mhz \*", _read(os.path.join(dev, "pp_dpm_sclk")) or "")
        mhz = float(m.group(1)) if m else None
    return (mhz, w) if (w is not None and mhz is not None) else None


class PowerSampler(threading.Thread):
    def __init__(self, ordinals, pci_of_ordinal=None):
        """ordinals: HIP device ordinals of the job's ranks (rank r uses ordinals[r]); pci_of_ordinal: {ordinal: 'dddd:bb:dd.f'} when known"""
        super().__init__(daemon=True)
        self.ordinals = list(ordinals)
        self.samples = []                 # (t, [(mhz, w) or None per ordinal])
        self.stop_flag, self.cap, self.source = threading.Event(), None, None
        cards = amdgpu_cards()
        self.paths = None
        if cards:
            by_pci = {c[0].lower(): c for c in cards}
            sel = []
            for o in self.ordinals:
                c = by_pci.get((pci_of_ordinal or {}).get(o, "").lower())
                if c is None and o < len(cards):
                    c = cards[o]          # (HIP enumerates in PCI order unless the visible-devices variables reorder)
                sel.append(c)
            if all(c is not None for c in sel) and _card_sample(sel[0][1], sel[0][2]) is not None:
                self.paths, self.source = sel, "sysfs hwmon (power1_average / freq1_input), read by rank 0 for every GPU of the job"
                cap = _read(os.path.join(sel[0][1], "power1_cap"))
                self.cap = int(cap) * 1e-6 if cap and cap.isdigit() else None
        vis = os.environ.get("HIP_VISIBLE_DEVICES") or os.environ.get("ROCR_VISIBLE_DEVICES")
        try:
            self.smi_idx = [int(vis.split(",")[o]) if vis else o for o in self.ordinals]
        except Exception:
            self.smi_idx = self.ordinals

    def _smi_all(self):
        """one rocm-smi child for all devices: {index: (mhz, w)}"""
        r = subprocess.run(["rocm-smi", "--showpower", "--showclocks"], capture_output=True, text=True, timeout=15)
        mhz = {int(a): float(b) for a, b in re.findall(r"GPU\[(\d+)\]\s*: sclk clock level: \S+ \((\d+)Mhz\)", r.stdout)}
        w = {int(a): float(b) for a, b in re.findall(r"GPU\[(\d+)\]\s*: [\w ]*Package Power \(W\): ([\d.]+)", r.stdout)}
        return {i: (mhz[i], w[i]) for i in mhz if i in w}

    def run(self):
        try:
            if self.paths is None:
                self.source = "rocm-smi (one child process per second for all devices, started by rank 0)"
                r = subprocess.run(["rocm-smi", "-d", str(self.smi_idx[0]), "--showmaxpower"], capture_output=True, text=True, timeout=15)
                m = re.search(r"Max Graphics Package Power \(W\): ([\d.]+)", r.stdout)
                self.cap = float(m.group(1)) if m else None
            while not self.stop_flag.wait(1.0):
                t = time.perf_counter()
                if self.paths is not None:
                    row = [_card_sample(c[1], c[2]) for c in self.paths]
                else:
                    got = self._smi_all()
                    row = [got.get(i) for i in self.smi_idx]
                self.samples.append((t, row))
        except Exception:
            pass

    def summary(self, t0, t1):
        self.stop_flag.set()
        sel = [row for t, row in self.samples if t0 + 1.0 <= t <= t1]
        per = []
        for k in range(len(self.ordinals)):
            v = [row[k] for row in sel if row[k] is not None]
            per.append((float(np.median([a for a, _ in v])), float(np.median([b for _, b in v]))) if v else None)
        if not per or per[0] is None:
            return dict({"package_power_w_median": None, "sclk_mhz_median": None, "power_cap_w": self.cap, "samples": 0, "source": self.source,
                         "note": "no sample inside the timed region"}, **({"over_gpus": None} if len(self.ordinals) > 1 else {}))
        out = {"package_power_w_median": per[0][1], "sclk_mhz_median": per[0][0], "power_cap_w": self.cap, "samples": len(sel), "source": self.source,
               "note": "rank 0's GPU, one sample per second inside the timed region (performance level auto): at the cap the firmware lowers the shader "
                       "clock (2.4 GHz maximum) until the package fits — solves/s = cap / energy per solve"}
        if len(per) > 1:
            have = [p for p in per if p is not None]
            out["over_gpus"] = {"gpus_sampled": len(have), "sclk_mhz_median_min": min(p[0] for p in have), "sclk_mhz_median_max": max(p[0] for p in have),
                                "package_power_w_median_min": min(p[1] for p in have), "package_power_w_median_max": max(p[1] for p in have),
                                "package_power_w_sum": sum(p[1] for p in have),
                                "note": "medians of every GPU of the job over the timed region: a node-level power or thermal limit shows as clocks below the lone-GPU "
                                        "figure on ALL of them at packages below their own caps"}
        return out
