NAME          RAW3COST
ROWS
 N  COST
 G  SUP1COST
 G  SUP2COST
 L  PURITY
 E  AMOUNT
COLUMNS
    SUP1      COST              1.40   SUP1COST          1.40
    SUP1      PURITY             .01   AMOUNT            1.00
    SUP2      COST               .70   SUP2COST           .70
    SUP2      PURITY             .07   AMOUNT            1.00
RHS
    RHS       AMOUNT          250.00   PURITY           12.50
BOUNDS
 UP BOUND     SUP2            150.00
ENDATA