* PULSE / PWL sources (superset of the reference dialect): CMOS inverter driven by a
* periodic pulse, RC on a PWL source with a DC term, RC on a single-shot current pulse
VDD 103 0 DC 3
VIN 101 0 PULSE(0 3 1n 0.5n 0.5n 2n 6n)
VB  110 0 DC 0.2 PWL 0 0 2n 1 5n 0.5 9n 2
I1  0 120 PULSE 0 1m 2n 1n 1n 3n
R1 101 102 1k
C1 102 0 0.1p
M1 104 102 103 p 30e-6 0.35e-6 1
M2 104 102 0   n 10e-6 0.35e-6 2
R2 104 0 100k
C2 104 0 0.05p
R3 110 111 2k
C3 111 0 0.2p
R4 120 0 1k
C4 120 0 1p
.MODEL 1 VT -0.75 MU 5e-2 COX 0.3e-4 LAMBDA 0.05 CJ0 4.0e-14
.MODEL 2 VT 0.83 MU 1.5e-1 COX 0.3e-4 LAMBDA 0.05 CJ0 4.0e-14
.TRAN 0.05n 20n
.PRINT TRAN V(104) V(111) V(120)
